// Shader-clock probe (measurement aid for tools/bench_ramp.py): one wavefront reads the core-clock counter (s_memtime,
// clock64) and the constant 100 MHz counter (s_memrealtime, wall_clock64) across a fixed ~10 us spin, so that
// sclk [MHz] = 100 * d(clock64) / d(wall_clock64) at the moment the kernel ran.  No privileges, no SMI.
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probe/clock_probe.hip -o tools/probe/libclock_probe.so
#include <hip/hip_runtime.h>

__global__ void k_clock_probe(long long *out, int slot, int spin_ticks) {
  if (threadIdx.x != 0) return;
  const long long w0 = wall_clock64();
  const long long c0 = clock64();
  long long w1 = w0;
  while (w1 - w0 < spin_ticks) w1 = wall_clock64();
  const long long c1 = clock64();
  out[2 * slot] = c1 - c0;
  out[2 * slot + 1] = w1 - w0;
}

extern "C" int clock_probe_launch(void *stream, long long *out, int slot, int spin_ticks) {
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, out, slot, spin_ticks);
  return (int)hipGetLastError();
}
