// How many wave64 vector instructions per second does an MI355X issue?  (tools/experiments: the number DESIGN.md section 4.3 prices
// the megakernel's VALU rate against.)  One wave per block, W waves per SIMD, each wave runs ITERS x 16 independent instructions of one
// kind.   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 8192
#define REP16(X) X X X X X X X X X X X X X X X X
template<int KIND>
__global__ __launch_bounds__(64) void k(float *out, float a, float b, int nact) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, pa = {a, a}, pb = {b, b};
    if ((int) threadIdx.x < nact)
    for (int i = 0; i < ITERS; ++i) {
        if (KIND == 0) { // v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (KIND == 1) { // v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        } else if (KIND == 2) { // v_min_f32 / v_max_f32 / v_add_u32 / v_cndmask mix (the integer / select work of a traversal step)
            asm volatile("v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_add_u32 %2, %2, %3\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_add_u32 %6, %6, %7\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         "v_min_f32 %0, %0, %8\n v_max_f32 %1, %1, %9\n v_add_u32 %2, %2, %3\n v_cndmask_b32 %3, %3, %4, vcc\n"
                         "v_min_f32 %4, %4, %8\n v_max_f32 %5, %5, %9\n v_add_u32 %6, %6, %7\n v_cndmask_b32 %7, %7, %0, vcc\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
        } else { // quad-permute DPP moves
            asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template<int KIND>
static void run(const char *name, int waves_per_simd, float *out, int nact = 64) {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int simds = pr.multiProcessorCount * 4, blocks = simds * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1.0000001f, 1e-9f, nact);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1.0000001f, 1e-9f, nact);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double) blocks * ITERS * 16.0, rate = instr / (ms * 1e-3);
    printf("%-12s %2d lanes %d waves/SIMD: %7.3f ms, %.3e wave-instructions/s, %.2f cycles per instruction per SIMD at %d MHz (%d SIMDs)\n", name, nact, waves_per_simd, ms, rate,
           (double) simds * pr.clockRate * 1e3 / rate, pr.clockRate / 1000, simds);
}
int main() {
    float *out; hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float) * 2);
    for (int w : {1, 2, 4, 6, 8}) { run<0>("v_fma_f32", w, out); }
    for (int w : {1, 2, 6}) { run<1>("v_pk_fma_f32", w, out); }
    for (int w : {1, 2, 6}) { run<2>("min/max/add/cndmask", w, out); }
    for (int w : {1, 2, 6}) { run<3>("v_mov_dpp", w, out); }
    // does the SIMD skip the 16-lane passes of a wave64 instruction whose EXEC bits are all zero?  (lanes >= nact are masked off)
    for (int n : {48, 32, 16, 4}) { run<2>("min/max/add/cndmask", 6, out, n); }
    for (int n : {32, 16}) { run<0>("v_fma_f32", 6, out, n); }
    return 0;
}
