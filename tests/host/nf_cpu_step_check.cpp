// Sanitizer driver of oracle/nf_cpu_step.cpp (TEST INFRASTRUCTURE; SURVEY.md section 5: the reference's CI has no
// native code to sanitise, this build has): built by tests/test_sanitizers.py with -fsanitize=address,undefined and
// run on inputs the test writes; the test compares the outputs with oracle/nf_oracle.py.
//   usage: nf_cpu_step_check <in.bin> <out.bin>
//   in : int32 d, h1, h2, nlayers, n ; float theta[P], mu[d], var[d], xs[n * d]  (xs sample-major)
//   out: float loss, gnorm, grad[P]   -- first for the supplied xs, then a second record for in-library draws + one Adam step
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../oracle/nf_cpu_step.cpp"

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  int hdr[5];
  if (std::fread(hdr, sizeof(int), 5, f) != 5) return 4;
  const int d = hdr[0], h1 = hdr[1], h2 = hdr[2], nl = hdr[3], n = hdr[4];
  const long P = nfcpu_param_count(d, h1, h2, nl);
  std::vector<float> theta(P), mu(d), var(d), xs((size_t)n * d), grad(P), m(P, 0.f), v(P, 0.f);
  if (std::fread(theta.data(), 4, P, f) != (size_t)P || std::fread(mu.data(), 4, d, f) != (size_t)d ||
      std::fread(var.data(), 4, d, f) != (size_t)d || std::fread(xs.data(), 4, (size_t)n * d, f) != (size_t)n * d)
    return 5;
  std::fclose(f);
  FILE *o = std::fopen(argv[2], "wb");
  if (!o) return 6;
  float loss = 0.f, gn = 0.f;
  // (i) value and gradient on the supplied draws, theta untouched (lr = 0, no optimiser state)
  std::vector<float> th = theta;
  int rc = nfcpu_realnvp_step(d, h1, h2, nl, th.data(), nullptr, nullptr, mu.data(), var.data(), xs.data(), n, 0, 1, 0.0f, &loss, &gn,
                              grad.data(), 3);
  if (rc != 0) return 10 + rc;
  std::fwrite(&loss, 4, 1, o);
  std::fwrite(&gn, 4, 1, o);
  std::fwrite(grad.data(), 4, P, o);
  // (ii) a full training step with in-library draws and Adam
  th = theta;
  rc = nfcpu_realnvp_step(d, h1, h2, nl, th.data(), m.data(), v.data(), mu.data(), var.data(), nullptr, n + 64, 123, 1, 1e-3f, &loss, &gn,
                          grad.data(), 4);
  if (rc != 0) return 20 + rc;
  std::fwrite(&loss, 4, 1, o);
  std::fwrite(&gn, 4, 1, o);
  std::fwrite(th.data(), 4, P, o);
  std::fclose(o);
  return 0;
}
