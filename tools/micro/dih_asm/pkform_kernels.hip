// Which FORM of a packed-FP32 instruction is the vulnerable one?  One victim per operand-select form of v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (inline asm, 16 instructions per
// set), for partner_classes.cpp.   hipcc --genco --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pkform.hsaco pkform_kernels.hip
// (scalar code around the asm statement and no SLP vectoriser: the inline-asm instruction is the only packed one of an iteration)
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int SETS = 16;
#define VICTIM(NAME, INSTR)                                                                                           \
    extern "C" __global__ __launch_bounds__(256) void NAME(const float4* __restrict__ pts, float* __restrict__ out, int n) { \
        const int t = blockIdx.x * 256 + threadIdx.x;                                                                 \
        for (int s = 0; s < SETS; ++s) {                                                                              \
            const float4 b = pts[((t * SETS + s) % n) * 4];                                                           \
            f32x2 x = {b.x * 0.01f + 1.f, b.y * 0.01f + 1.f}, y = {b.z * 0.01f + 1.f, b.x * 0.02f + 1.f};             \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                                          \
                f32x2 r;                                                                                              \
                asm volatile(INSTR : "=v"(r) : "v"(x), "v"(y));                                                       \
                x = y; y[0] = __builtin_fmaf(r[0], 0.5f, 0.25f); y[1] = __builtin_fmaf(r[1], 0.5f, 0.25f);                                                                          \
            }                                                                                                         \
            out[(size_t)(t * SETS + s) * 2 + 0] = x[0] + y[0];                                                        \
            out[(size_t)(t * SETS + s) * 2 + 1] = x[1] + y[1];                                                        \
        }                                                                                                             \
    }
VICTIM(pk_mul_plain, "v_pk_mul_f32 %0, %1, %2")
VICTIM(pk_mul_bcast_lo, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]")
VICTIM(pk_mul_bcast_hi, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]")
VICTIM(pk_mul_cross, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]")
VICTIM(pk_add_plain, "v_pk_add_f32 %0, %1, %2")
VICTIM(pk_add_cross, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]")
VICTIM(pk_add_neg, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")
VICTIM(pk_fma_plain, "v_pk_fma_f32 %0, %1, %2, %1")
