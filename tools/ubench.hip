// ubench.hip -- VALU integer instruction-rate probe for gfx950 (tooling, not product).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run on the GPU box.
// Reports wave-instructions per cycle per SIMD-equivalent for the ops the Goldilocks/Poseidon
// kernels are built from, so the kernel design can be priced against measured rates.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define ITERS 4096
#define NACC 8

template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t *out, uint32_t seed) {
    uint32_t t = threadIdx.x + blockIdx.x * blockDim.x;
    uint64_t a[NACC];
    uint32_t b[NACC], c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        a[i] = (uint64_t)t * 0x9E3779B97F4A7C15ull + i + seed;
        b[i] = t * 2654435761u + i * 7 + seed;
        c[i] = (t ^ seed) + i * 13 + 1;
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (OP == 0) {  // v_mad_u64_u32
                a[i] = (uint64_t)b[i] * c[i] + a[i];
                b[i] = (uint32_t)a[i];
            } else if (OP == 1) {  // v_mul_lo_u32
                b[i] = b[i] * c[i];
            } else if (OP == 2) {  // v_mul_hi_u32
                b[i] = __umulhi(b[i], c[i]);
            } else if (OP == 3) {  // v_mad_u32_u24
                b[i] = (b[i] & 0xFFFFFFu) * (c[i] & 0xFFFFFFu) + b[i];
            } else if (OP == 4) {  // v_add_u32
                b[i] = b[i] + c[i];
            } else if (OP == 5) {  // 64-bit add (add_co + addc)
                a[i] = a[i] + (((uint64_t)c[i] << 32) | b[i]);
            } else if (OP == 6) {  // v_mul_u32_u24 with inline constant
                b[i] = (b[i] & 0x3FFFFFu) * 41u + c[i];
            } else if (OP == 7) {  // v_lshl_add_u64-ish
                a[i] = (a[i] << 3) + (uint64_t)c[i];
            } else if (OP == 8) {  // v_alignbit / shifts+and (limb split)
                b[i] = ((b[i] >> 22) | (c[i] << 10)) & 0x1FFFFFu;
                c[i] = c[i] + b[i];
            }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += a[i] + b[i] + c[i];
    out[t] = s;
}

template <int OP>
void run(const char *name, double insts_per_iter) {
    const int blocks = 256 * 8, threads = 256;
    uint64_t *d;
    hipMalloc(&d, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(d, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)blocks * threads / 64;
    double winst = waves * ITERS * NACC * insts_per_iter;
    // 1024 SIMDs at 2.4 GHz
    printf("%-28s %8.3f ms  %7.2f Gwave-inst/s  = %.3f wave-inst/clk/SIMD (@2.4GHz,1024 SIMDs)\n", name, ms, winst / ms / 1e6,
           winst / (ms * 1e-3) / (1024 * 2.4e9));
    hipFree(d);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs %d  clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    run<4>("v_add_u32", 1);
    run<0>("v_mad_u64_u32 (+mov)", 1);
    run<1>("v_mul_lo_u32", 1);
    run<2>("v_mul_hi_u32", 1);
    run<3>("v_mad_u32_u24 (+2 and)", 1);
    run<6>("v_mul_u32_u24 imm (+and,add)", 1);
    run<5>("add u64 (2 insts)", 1);
    run<7>("shl3+add u64", 1);
    run<8>("limb split (3-4 insts)", 1);
    return 0;
}
