// Host-only sanitizer check of nf_api.hip's argument validation, workspace sizing and arena carving
// (TEST INFRASTRUCTURE; SURVEY.md section 5 / ADVICE r1-r2: a misaligned carve and a tail carve into the live front were
// both found by reading -- this finds that class mechanically).  Built by tests/test_sanitizers.py:
//   hipcc --offload-host-only -fsanitize=address,undefined -O1 -g   (every translation unit of the library, host side only)
// and run WITHOUT a GPU: no kernel is launched and no HIP call is made; the context is a hand-made struct whose arena
// is a fake device address that is never dereferenced (the carving code only does pointer arithmetic on it).
// This TU includes nf_api.hip to reach its file-local functions.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../normalizingflows.jl_amd/csrc/nf_api.hip"

static int fails = 0;
#define CHECK(cond)                                                              \
  do {                                                                           \
    if (!(cond)) {                                                               \
      std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond);    \
      ++fails;                                                                   \
    }                                                                            \
  } while (0)

static nf_flow_desc mk(int kind, int dtype, int d, int nl, std::initializer_list<int> hd = {}, int K = 0, float B = 0.f) {
  nf_flow_desc g;
  std::memset(&g, 0, sizeof g);
  g.kind = kind;
  g.dtype = dtype;
  g.d = d;
  g.nlayers = nl;
  g.n_hidden = (int)hd.size();
  int i = 0;
  for (int h : hd) g.hdims[i++] = h;
  g.K = K;
  g.B = B;
  return g;
}

static long mlp(int nin, std::initializer_list<int> hd, int nout) {
  long n = 0;
  int prev = nin;
  for (int h : hd) {
    n += (long)prev * h + h;
    prev = h;
  }
  return n + (long)prev * nout + nout;
}

int main() {
  // ---- parameter counts against the closed forms of Optimisers.destructure ------------------------
  {
    nf_flow_desc a = mk(NF_KIND_REALNVP, NF_DTYPE_F32, 64, 4, {64, 64});
    CHECK(nf_param_count(&a) == 8 * 2 * mlp(32, {64, 64}, 32));
    CHECK(nf_param_count(&a) == 133632);  // cfg 2
    nf_flow_desc b = mk(NF_KIND_NSF, NF_DTYPE_F32, 32, 4, {32, 32}, 8, 5.f);
    CHECK(nf_param_count(&b) == 8 * mlp(16, {32, 32}, 23 * 16));
    CHECK(nf_param_count(&b) == 109952);  // cfg 3
    nf_flow_desc c = mk(NF_KIND_REALNVP, NF_DTYPE_F32, 5, 2, {32, 32});  // odd d: masks of unequal size (test/flow.jl:4)
    CHECK(nf_param_count(&c) == 2 * (2 * mlp(2, {32, 32}, 3) + 2 * mlp(3, {32, 32}, 2)));
    nf_flow_desc p = mk(NF_KIND_PLANAR, NF_DTYPE_F64, 2, 10);
    CHECK(nf_param_count(&p) == 50);  // cfg 1
    CHECK(nf_layer_count(&a) == 8 && nf_layer_count(&p) == 10);
    nf_flow_desc w = mk(NF_KIND_REALNVP, NF_DTYPE_F32, 256, 8, {256, 256});
    CHECK(nf_param_count(&w) == 4214784);  // cfg 4
  }
  // ---- descriptor validation: bad shapes are refused with the documented codes, never accepted -----------------------
  {
    CHECK(check_desc(nullptr) == NF_ERR_ARG);
    nf_flow_desc g = mk(NF_KIND_REALNVP, NF_DTYPE_F32, 64, 4, {64, 64});
    CHECK(check_desc(&g) == NF_OK);
    g.d = 0;
    CHECK(check_desc(&g) == NF_ERR_ARG);
    g = mk(NF_KIND_REALNVP, 7, 64, 4, {64, 64});
    CHECK(check_desc(&g) == NF_ERR_ARG);
    g = mk(NF_KIND_REALNVP, NF_DTYPE_F32, 64, 0, {64, 64});
    CHECK(check_desc(&g) == NF_ERR_ARG);
    g = mk(NF_KIND_REALNVP, NF_DTYPE_F32, 600, 1, {32, 32});
    CHECK(check_desc(&g) == NF_ERR_UNSUPPORTED);
    g = mk(NF_KIND_REALNVP, NF_DTYPE_F32, 16, 1, {512, 512});
    CHECK(check_desc(&g) == NF_ERR_UNSUPPORTED);
    g = mk(NF_KIND_NSF, NF_DTYPE_F32, 8, 1, {32, 32}, 1, 5.f);
    CHECK(check_desc(&g) == NF_ERR_ARG);  // K < 2
    g = mk(99, NF_DTYPE_F32, 8, 1);
    CHECK(check_desc(&g) == NF_ERR_ARG);
    g = mk(NF_KIND_HAMILTONIAN, NF_DTYPE_F64, 5, 1, {}, 3);
    CHECK(check_desc(&g) == NF_ERR_ARG);  // odd joint dimension, no score target
    nf_base bad;
    bad.kind = 9;
    bad.mu = bad.scale = nullptr;
    bad.logdet = 0;
    g = mk(NF_KIND_PLANAR, NF_DTYPE_F32, 4, 2);
    g.base = &bad;
    CHECK(check_desc(&g) == NF_ERR_ARG);
    // compositions: segments must share d / dtype, carry no base of their own, and not nest
    nf_flow_desc segs[3] = {mk(NF_KIND_PLANAR, NF_DTYPE_F32, 8, 2), mk(NF_KIND_REALNVP, NF_DTYPE_F32, 8, 1, {32, 32}),
                            mk(NF_KIND_RADIAL, NF_DTYPE_F32, 8, 3)};
    nf_flow_desc c = mk(NF_KIND_COMPOSITE, NF_DTYPE_F32, 8, 1);
    c.nsegments = 3;
    c.segments = segs;
    CHECK(check_desc(&c) == NF_OK);
    CHECK(nf_param_count(&c) == nf_param_count(&segs[0]) + nf_param_count(&segs[1]) + nf_param_count(&segs[2]));
    CHECK(nf_layer_count(&c) == 2 + 2 + 3);
    segs[1].d = 6;
    CHECK(check_desc(&c) == NF_ERR_ARG);
    segs[1].d = 8;
    segs[2].dtype = NF_DTYPE_F64;
    CHECK(check_desc(&c) == NF_ERR_ARG);
    segs[2].dtype = NF_DTYPE_F32;
    c.nsegments = 0;
    CHECK(check_desc(&c) == NF_ERR_ARG);
    c.nsegments = 65;
    CHECK(check_desc(&c) == NF_ERR_ARG);
  }
  // ---- sizing: one bound covers every entry point; arena carving stays inside it and never overlaps -------------------
  nf_ctx ctx;  // hand-made: no device, no stream, nothing is launched
  ctx.num_cu = 256;
  std::vector<nf_flow_desc> flows = {
      mk(NF_KIND_REALNVP, NF_DTYPE_F32, 64, 4, {64, 64}),   mk(NF_KIND_REALNVP, NF_DTYPE_F32, 20, 2, {32, 32}),
      mk(NF_KIND_REALNVP, NF_DTYPE_F32, 63, 2, {40, 64}),   mk(NF_KIND_REALNVP, NF_DTYPE_F32, 256, 8, {256, 256}),
      mk(NF_KIND_REALNVP, NF_DTYPE_F32, 100, 2, {128, 96}), mk(NF_KIND_NSF, NF_DTYPE_F32, 32, 4, {32, 32}, 8, 5.f),
      mk(NF_KIND_NSF, NF_DTYPE_F32, 5, 2, {32, 32}, 10, 5.f), mk(NF_KIND_NSF, NF_DTYPE_F32, 32, 2, {64, 64}, 8, 3.f),
      mk(NF_KIND_NSF, NF_DTYPE_F64, 6, 1, {24, 16, 8}, 8, 5.f), mk(NF_KIND_REALNVP, NF_DTYPE_F64, 5, 2, {32, 32}),
      mk(NF_KIND_REALNVP, NF_DTYPE_F32, 9, 1, {24, 16, 8}), mk(NF_KIND_PLANAR, NF_DTYPE_F32, 64, 10),
      mk(NF_KIND_PLANAR, NF_DTYPE_F64, 2, 10),              mk(NF_KIND_RADIAL, NF_DTYPE_F32, 5, 10),
      mk(NF_KIND_MEANFIELD, NF_DTYPE_F64, 4, 1),            mk(NF_KIND_PLANAR, NF_DTYPE_F32, 200, 30)};
  nf_flow_desc csegs[3] = {mk(NF_KIND_PLANAR, NF_DTYPE_F32, 64, 3), mk(NF_KIND_REALNVP, NF_DTYPE_F32, 64, 2, {64, 64}),
                           mk(NF_KIND_NSF, NF_DTYPE_F32, 64, 1, {32, 32}, 8, 5.f)};
  nf_flow_desc comp = mk(NF_KIND_COMPOSITE, NF_DTYPE_F32, 64, 1);
  comp.nsegments = 3;
  comp.segments = csegs;
  flows.push_back(comp);
  const long Ns[] = {1, 31, 32, 33, 1000, 4096 + 17, 65536};
  for (long budget : {-1L, 0L, 3L << 20}) {
    ctx.stash_budget = budget;
    for (const nf_flow_desc &g : flows) {
      if (check_desc(&g) != NF_OK) {
        CHECK(check_desc(&g) == NF_ERR_UNSUPPORTED);
        continue;
      }
      int64_t prev = 0;
      for (long N : Ns) {
        const int64_t total = nf_workspace_bytes(&ctx, &g, N);
        CHECK(total > 0 && total % 256 == 0);
        CHECK(total >= prev);  // monotone in N
        prev = total;
        const size_t bound = ws_need_bound(&ctx, &g, N);
        CHECK(bound % 256 == 0);
        CHECK(bound >= flow_bwd_need(&ctx, &g, N));
        CHECK(bound >= tape_need(&ctx, &g, N, true) && bound >= tape_need(&ctx, &g, N, false));
        if (is_composite(&g)) CHECK(bound >= vg_composite_need(&ctx, &g, N));
        const int64_t tb = nf_tape_bytes(&ctx, &g, N);
        CHECK(tb > 0 && tb % 256 == 0);
        if (is_composite(&g)) {
          int64_t s = 0;
          for (int i = 0; i < g.nsegments; ++i) s += nf_tape_bytes(&ctx, &g.segments[i], N);
          CHECK(s == tb);
        }
        if (g.kind == NF_KIND_REALNVP && g.dtype == NF_DTYPE_F32 && nf_affine_supported(&g)) {
          const long chunk = affine_stash_chunk(&ctx, &g, N);
          if (budget == 0) CHECK(chunk == 0);
          if (chunk) {
            CHECK(chunk == N || chunk % 32 == 0);
            const size_t sb = affine_stash_bytes(&ctx, &g, chunk);
            CHECK(sb > 0 && (budget <= 0 || sb <= (size_t)budget));
            CHECK((size_t)total >= step_fused_need(&ctx, &g, N) + carve_bytes(nf_affine_wimg_bytes(&g)));  // nf_elbo_step's own form
            // slabs of a chunked step: never more than one slab per workgroup of every chunk
            const long stride = coupling_slab_floats(&ctx, &g, N);
            const size_t sf = chunked_slab_floats(&ctx, &g, N, chunk, stride);
            const long nch = (N + chunk - 1) / chunk;
            CHECK(sf <= (size_t)nch * coupling_bwd_grid(&ctx, &g, chunk) * stride);
            CHECK(sf >= (size_t)coupling_bwd_grid(&ctx, &g, N < chunk ? N : chunk) * stride);
          }
        }
        // arena mode: the carving of an entry point (front intermediates, then tail carves) inside exactly `total` bytes
        char *fake = (char *)(uintptr_t)0x7f0000000000ull;  // never dereferenced
        ctx.arena = fake;
        ctx.arena_bytes = (size_t)total;
        ctx.arena_tail = ctx.arena_front = 0;
        ctx.ws = ctx.wimg = ctx.gbuf = nullptr;
        ctx.ws_bytes = ctx.wimg_bytes = ctx.gbuf_bytes = 0;
        CHECK(nf_ws_reserve(&ctx, bound) == NF_OK);
        CHECK(ctx.ws == (void *)fake && ctx.arena_front == bound);
        Carver cv(ctx.ws);
        char *a0 = cv.take<char>(100), *a1 = cv.take<float>(7) ? (char *)cv.base + cv.off : nullptr;
        CHECK(((uintptr_t)a0 & 255) == 0 && ((uintptr_t)a1 & 255) == 0);
        size_t wimg = 0;
        if (is_coupling(&g)) wimg = is_wide(&g) ? nf_wide_wimg_bytes(&ctx, &g) : is_nsf(&g) ? nf_rqs_wimg_bytes(&g) : nf_affine_wimg_bytes(&g);
        if (wimg) {
          CHECK(nf_wimg_reserve(&ctx, wimg) == NF_OK);
          CHECK((char *)ctx.wimg >= fake + bound);                              // behind the live front
          CHECK((char *)ctx.wimg + wimg <= fake + total);                       // inside the arena
          CHECK(((uintptr_t)ctx.wimg & 255) == 0);
        }
        void *gb = nullptr;
        const size_t gneed = gbuf_need(nf_param_count(&g), esize(g.dtype));
        CHECK(arena_tail_take(&ctx, gneed, &gb) == NF_OK);
        CHECK((char *)gb >= fake + bound && (char *)gb + gneed <= (wimg ? (char *)ctx.wimg : fake + total));
        // an arena that holds the front but not the tails must be REFUSED (round 2 overlapped them silently)
        ctx.arena_bytes = carve_bytes(bound) + 256;
        ctx.arena_tail = ctx.arena_front = 0;
        ctx.wimg = nullptr;
        ctx.wimg_bytes = 0;
        CHECK(nf_ws_reserve(&ctx, bound) == NF_OK);
        if (wimg > 256) CHECK(nf_wimg_reserve(&ctx, wimg) == NF_ERR_WORKSPACE);
        CHECK(arena_tail_take(&ctx, gneed > 512 ? gneed : 512, &gb) == NF_ERR_WORKSPACE);
        ctx.arena_bytes = bound > 512 ? bound - 256 : 0;
        ctx.arena_tail = ctx.arena_front = 0;
        if (ctx.arena_bytes) CHECK(nf_ws_reserve(&ctx, bound) == NF_ERR_WORKSPACE);
        // a wrapper's guard: inner requests beyond it are refused
        ctx.arena_bytes = (size_t)total;
        ctx.arena_tail = ctx.arena_front = 0;
        CHECK(nf_ws_reserve(&ctx, bound) == NF_OK);
        ctx.ws_guard = 4096;
        CHECK(nf_ws_reserve(&ctx, 8192) == NF_ERR_WORKSPACE && nf_ws_reserve(&ctx, 4096) == NF_OK);
        if (bound > 4096) CHECK(ctx.arena_front == bound);  // the outer request stays the mark while a guard is set
        ctx.ws_guard = 0;
        ctx.arena = nullptr;
        ctx.ws = ctx.wimg = ctx.gbuf = nullptr;
        ctx.ws_bytes = ctx.wimg_bytes = ctx.gbuf_bytes = 0;
      }
    }
  }
  // ---- null / range checks of public entry points that must fail before touching a device -----------------------------
  {
    nf_flow_desc g = mk(NF_KIND_REALNVP, NF_DTYPE_F32, 64, 4, {64, 64});
    float dummy = 0.f;
    CHECK(nf_flow_fwd(nullptr, &g, &dummy, &dummy, 1, &dummy, &dummy) == NF_ERR_ARG);
    CHECK(nf_flow_fwd(&ctx, &g, nullptr, &dummy, 1, &dummy, &dummy) == NF_ERR_ARG);
    CHECK(nf_flow_fwd(&ctx, &g, &dummy, &dummy, -1, &dummy, &dummy) == NF_ERR_ARG);
    CHECK(nf_flow_fwd_keep(&ctx, &g, &dummy, &dummy, 1, &dummy, &dummy, nullptr, 0) == NF_ERR_ARG);
    CHECK(nf_flow_fwd_keep(&ctx, &g, &dummy, &dummy, 1, &dummy, &dummy, (void *)(uintptr_t)0x1001, 1 << 20) == NF_ERR_ARG);  // unaligned tape
    CHECK(nf_elbo_step(&ctx, &g, nullptr, &dummy, &dummy, &dummy, 16, 0, 0, 1e-3, 0.9, 0.999, 1e-8, nullptr, nullptr) == NF_ERR_ARG);
    CHECK(nf_adam_update(&ctx, NF_DTYPE_F32, &dummy, &dummy, &dummy, &dummy, 4, 1e-3, 0.9, 0.999, 1e-8, 0, nullptr) == NF_ERR_ARG);  // t >= 1
    CHECK(nf_ctx_set_arena(&ctx, (void *)(uintptr_t)0x1010, 1 << 20) == NF_ERR_ARG);  // misaligned arena
    CHECK(nf_tape_bytes(&ctx, &g, -1) == NF_ERR_ARG && nf_workspace_bytes(nullptr, &g, 1) == NF_ERR_ARG);
    CHECK(std::string(nf_strerror(NF_ERR_WORKSPACE)).find("arena") != std::string::npos);
  }
  if (fails) {
    std::fprintf(stderr, "%d check(s) failed\n", fails);
    return 1;
  }
  std::printf("nf_api host check: ok\n");
  return 0;
}
