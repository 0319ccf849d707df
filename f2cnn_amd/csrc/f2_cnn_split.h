// Split-FP16 arithmetic of K4's matrix kernels (f2_cnn.hip: one workgroup per tile; f2_cnn_ws.hip: weight-stationary).
// Reference: the float32 network of scripts/CNN/Training.py:93-114 as keras model.predict evaluates it (Evaluating.py:85-87).
//
// v_mfma_f32_32x32x16_f16 issues at 16 x the rate of v_mfma_f32_32x32x2_f32 (and at the rate of the bf16 form: tools/ubench/
// mfma_f16_split.hip, 1.86 PFLOP/s). With both GEMM operands split into two fp16 pieces,
//     a = a1 + a2,  a1 = fp16(a),  a2 = fp16(a - a1)        (11 + 11 significant bits; bf16 pieces, rounds 3-4: 8 + 8)
// the three products a1 b1 + a1 b2 + a2 b1 - each exact in float32, float32 accumulation - carry a product to ~2^-22: the scores
// come out at the float32 rounding level (1.2e-7 against 6.6e-7 with bf16 pieces, no label of the cfg4 corpus differs from the
// float32 oracle's: tests/diag/fp16_split_experiment.py, profiles/r05_fp16_split_experiment.txt).
//
// fp16's exponent range is narrow (normal from 6.1e-5, 65504 at the top; the matrix core keeps subnormal inputs, but they carry
// fewer bits), so every layer's operands are scaled by powers of two first - exact, and taken out of the accumulator again in the
// layer's epilogue, together with the next layer's scale:
//   activations entering layer L (conv2, conv3, conv4, dense1) are stored as a * sa_L, sa_L = the largest power of two that keeps
//     an UPPER BOUND of the layer's input (L1 norms of the weights of the layers before it, network inputs in [0, 1] as
//     normalizeInput leaves them - Training.py:13-28) at or below 2^14: inputs up to 4 cannot overflow, larger ones give inf / NaN
//     scores (the float32 kernels, option cnn_f16x3 = 0, take any range);
//   weights of layer L are stored as w * sb_L, sb_L = the power of two that puts max |w| into [2^10, 2^11);
//   the accumulator holds sa_L sb_L (w . a): the epilogue forms relu(acc * c_L + b_L * so_L) with c_L = so_L / (sa_L sb_L) and
//     so_L = the next layer's sa (1 after dense1) - one fma where there was an addition.
// conv1 (Cin = 1) has its weights scaled by sa_2 directly, so its outputs are conv2's scaled inputs.
#ifndef F2_CNN_SPLIT_H
#define F2_CNN_SPLIT_H

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// powers of two, computed once per network by f2_cnn_create (struct f2_cnn: sc)
struct f2_split_scales {
    float sa2;              // scale of conv1's weights and bias = scale of conv2's input
    float c2, c3, c4, cd;   // epilogue multipliers of conv2, conv3, conv4, dense1
    float sin_d;            // what the per-tile dense1 kernel multiplies its input by: 1 when conv4 left it scaled, sa_dense1 when
                            // it comes in true units (window shapes whose conv3 / conv4 run on the float32 kernels)
};

__device__ __forceinline__ void split_h16(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// Two values at a time: the hi pieces by one packed conversion, each lo piece by one mixed-precision FMA that reads its fp16
// hi piece straight from the packed register and rounds v - hi to fp16 into its half of the result (v_fma_mixlo / mixhi_f16):
// 3 instructions per pair where the compiler's rendering of the C expressions below takes 6-7 (it converts every hi piece back
// to float32 and subtracts there). Bit-identical to them over 8.4 M values incl. subnormal pieces (tools/ubench/split_mix_check.hip).
// Inline asm is invisible to the compiler's hazard recogniser: the operands must NOT be matrix-core results still in flight (every
// caller passes the output of a ReLU or a multiply, i.e. of a compiler-scheduled VALU instruction that already waited for them) -
// a v_max3 written this way straight on the accumulators read them too early and the create-time self-check switched the kernels off.
__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo) {
#ifdef F2_SPLIT_PLAIN    // (the C form, for A/B builds)
    typedef _Float16 h16x2_ __attribute__((ext_vector_type(2)));
    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
    hi = __builtin_bit_cast(unsigned, h16x2_{h0, h1});
    lo = __builtin_bit_cast(unsigned, h16x2_{(_Float16)(v0 - (float)h0), (_Float16)(v1 - (float)h1)});
#else
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
#endif
}

__device__ __forceinline__ void split4(const float v0, const float v1, const float v2, const float v3, h16x4& hi, h16x4& lo) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    u32x2_ h, l;
    unsigned a, b;
    split2(v0, v1, a, b);
    h[0] = a;
    l[0] = b;
    split2(v2, v3, a, b);
    h[1] = a;
    l[1] = b;
    hi = __builtin_bit_cast(h16x4, h);
    lo = __builtin_bit_cast(h16x4, l);
}

#endif
