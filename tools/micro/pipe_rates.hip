// Micro-benchmarks of the gfx950 pipes the wide solver leans on (diagnostic; GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pipe_rates.hip -o gpurun_out/pipe_rates && gpurun_out/pipe_rates
// Prints shader-clock ticks (s_memtime) per operation for: independent / dependent v_fma_f64,
// v_mfma_f64_16x16x4_f64 with 1 / 2 / 4 accumulators and with two wavefronts on a SIMD, the
// fma -> v_readlane -> fma chain of the triangular solves, an LDS round trip, a DPP wave
// reduction, and the ratio of s_memtime to the 100 MHz s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* t, int which) {
    __shared__ double lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (double)((i * 7 + 3) & 4095);
    __syncthreads();
    double a = 1.0 + lane * 1e-9, b = 0.999999, c0 = 0.1, c1 = 0.2, c2 = 0.3, c3 = 0.4, c4 = 0.5, c5 = .6, c6 = .7, c7 = .8;
    double4v C0 = {0, 0, 0, 0}, C1 = C0, C2 = C0, C3 = C0;
    unsigned long long t0 = 0, t1 = 0, w0 = 0, w1 = 0;
    const bool active = (which == 5) ? (wave == 0 || wave == 4) : (which == 6 ? true : wave == 0);
    __syncthreads();
    if (active) {
        w0 = wall_clock64();
        t0 = clock64();
        if (which == 0) {           // independent FMAs, 8 chains, 512 instructions
#pragma unroll 1
            for (int i = 0; i < 64; ++i) {
                c0 = fma(a, b, c0); c1 = fma(a, b, c1); c2 = fma(a, b, c2); c3 = fma(a, b, c3);
                c4 = fma(a, b, c4); c5 = fma(a, b, c5); c6 = fma(a, b, c6); c7 = fma(a, b, c7);
            }
        } else if (which == 1) {    // dependent chain, 512
#pragma unroll 8
            for (int i = 0; i < 512; ++i) c0 = fma(c0, b, a);
        } else if (which == 2) {    // MFMA, one accumulator, 128
#pragma unroll 4
            for (int i = 0; i < 128; ++i) C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C0, 0, 0, 0);
        } else if (which == 3) {    // two accumulators, 128
#pragma unroll 2
            for (int i = 0; i < 64; ++i) {
                C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C0, 0, 0, 0);
                C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C1, 0, 0, 0);
            }
        } else if (which == 4 || which == 5 || which == 6) {    // four accumulators, 128 per wave
#pragma unroll 2
            for (int i = 0; i < 32; ++i) {
                C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C0, 0, 0, 0);
                C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C1, 0, 0, 0);
                C2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C2, 0, 0, 0);
                C3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, C3, 0, 0, 0);
            }
        } else if (which == 7) {    // fma -> readlane -> fma chain, 256 steps
#pragma unroll 16
            for (int i = 0; i < 256; ++i) {
                const double y = readlane_d(c0, i & 63);
                c0 = fma(-b, y, c0);
            }
        } else if (which == 8) {    // dependent LDS loads, 128
            int idx = lane;
#pragma unroll 4
            for (int i = 0; i < 128; ++i) idx = (int)lds[idx & 4095];
            c0 = idx;
        } else if (which == 9) {    // 16 independent LDS loads then use, 32 rounds (512 loads)
            double s = 0;
#pragma unroll 1
            for (int i = 0; i < 32; ++i) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = lds[(lane + 64 * u + i) & 4095];
#pragma unroll
                for (int u = 0; u < 16; ++u) s += v[u];
            }
            c0 = s;
        } else if (which == 10) {   // DPP wave sum x 32
#pragma unroll 1
            for (int i = 0; i < 32; ++i) {
                double v = c0;
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                c0 = v * 1e-3;
            }
        } else if (which == 11) {   // workgroup barrier x 64 (all 8 waves must run: see launch)
        }
        t1 = clock64();
        w1 = wall_clock64();
    }
    if (which == 11) {
        __syncthreads();
        t0 = clock64();
#pragma unroll 1
        for (int i = 0; i < 64; ++i) __syncthreads();
        t1 = clock64();
    }
    double r = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    for (int q = 0; q < 4; ++q) r += C0[q] + C1[q] + C2[q] + C3[q];
    out[tid] = r;
    if (lane == 0) { t[2 * wave] = t1 - t0; t[2 * wave + 1] = w1 - w0; }
}
int main() {
    double* out; unsigned long long* t;
    hipMalloc(&out, 512 * 8); hipMalloc(&t, 16 * 8);
    const char* names[] = {"v_fma_f64 independent (per instr)", "v_fma_f64 dependent (per instr)",
        "mfma_f64_16x16x4, 1 accumulator", "mfma 2 accumulators", "mfma 4 accumulators",
        "mfma 4 acc, waves 0 and 4 (same SIMD?) per instr of a wave", "mfma 4 acc, all 8 waves, per instr of a wave",
        "fma -> readlane -> fma step", "dependent LDS load", "independent LDS load (16 in flight)", "wave sum by shuffles",
        "workgroup barrier (8 waves)"};
    const int counts[] = {512, 512, 128, 128, 128, 128, 128, 256, 128, 512, 32, 64};
    for (int which = 0; which < 12; ++which) {
        unsigned long long h[16];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, t, which);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
        printf("%-62s ticks/op: w0 %.1f", names[which], (double)h[0] / counts[which]);
        if (which == 5) printf("  w4 %.1f", (double)h[8] / counts[which]);
        if (which == 6) for (int w = 1; w < 8; ++w) printf(" w%d %.1f", w, (double)h[2 * w] / counts[which]);
        printf("   (s_memtime / s_memrealtime = %.2f)\n", h[1] ? (double)h[0] / h[1] : 0.0);
    }
    return 0;
}
