// Merged passes 1 + 2 of the 8192-point transform (see the comment below); included by f2_spectral.hip inside its
// anonymous namespace, after f2_fft_lds.h.
#pragma once
// ---- passes 1 and 2 of the 16-8-4-16 plan (H = 8192, 512 threads) in ONE trip through LDS ----
// Pass 2 (radix 4, stride 128) combines four outputs of pass 1 (radix 8, stride 16) that belong to butterflies
// b = q + 16 p, p = p_lo + 16 p_hi, differing in p_hi only. With pass 1's butterflies dealt so that p_hi = i + 2 (lane >> 5)
// (i = the thread's first / second butterfly) those four values sit in one lane pair (l, l + 32): eight
// v_permlane32_swap per butterfly pair give every lane all four inputs of four radix-4 butterflies (the low lane those
// of pass-1 outputs k = 0..3, the high lane k = 4..7), and the exchange pass 1 -> pass 2 (16 ds_write_b64 + 16 ds_read_b64
// per thread, two barriers) is gone. Same arithmetic, same twiddle tables, same result bits as fft_pass<1> + fft_pass<2>.
// Measured bound (a build that simply skips that exchange, results wrong): -5 % on the kernel.
template <int NT, int PTV>
__device__ __forceinline__ void fft13_pass12_merged(cpx<float>* lds, const cpx<float>* twl, int tid, cpx<float> (&v)[PTV]) {
    constexpr int LOG2H = 13;
    static_assert(NT == 512 && PTV >= 16 && plan_npass(LOG2H) == 4 && plan_bits(LOG2H, 1) == 3 && plan_bits(LOG2H, 2) == 2 &&
                      plan_bits(LOG2H, 0) == 4, "the 16-8-4-16 plan on 512 threads");
    constexpr int OFF2 = plan_tw_offset(LOG2H, 2) - plan_tw_offset(LOG2H, 1);   // pass 2's table inside twl
    const int l = tid & 63, hl = l >> 5;
    const int base8 = (l & 31) + 32 * (tid >> 6);        // q + 16 p_lo
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = base8 + 256 * (i + 2 * hl);
        const cpx<float>* src = lds + cpad(b);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i * 8 + j] = src[j * (1024 + 64)];   // cpad(b + 1024 j)
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        dft<8>(&v[i * 8]);                                  // X[k] in v[8 i + brev<8>(k)]
        const cpx<float>* t1 = twl + (((base8 >> 4) + 16 * (i + 2 * hl)));   // p = b >> 4
#pragma unroll
        for (int k = 1; k < 8; ++k) v[i * 8 + brev<8>(k)] = cmul(v[i * 8 + brev<8>(k)], t1[(k - 1) * 64]);
    }
    // lane pair exchange: A = (i, k), B = (i, k + 4), k < 4. Afterwards, in both lanes, A holds the value of p_hi = i and
    // B that of p_hi = i + 2 for the pass-1 output kk = k + 4 (lane >> 5)
    // (inline asm: this compiler mis-tracks the two results of __builtin_amdgcn_permlane32_swap once they are written
    // back over their inputs - it went on to read one register for both; the instruction updates both operands in
    // place, which "+v" states exactly. s_nop: no hazard handling inside an asm block.)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        cpx<float>& A0 = v[i * 8 + brev<8>(0)];
        cpx<float>& B0 = v[i * 8 + brev<8>(4)];
        cpx<float>& A1 = v[i * 8 + brev<8>(1)];
        cpx<float>& B1 = v[i * 8 + brev<8>(5)];
        cpx<float>& A2 = v[i * 8 + brev<8>(2)];
        cpx<float>& B2 = v[i * 8 + brev<8>(6)];
        cpx<float>& A3 = v[i * 8 + brev<8>(3)];
        cpx<float>& B3 = v[i * 8 + brev<8>(7)];
#ifdef F2_MERGE_SHFL   // diagnostic: the same exchange through ds_bpermute
        auto xchg = [&](cpx<float>& A, cpx<float>& B) {
            const cpx<float> pa = {__shfl_xor(A.re, 32), __shfl_xor(A.im, 32)}, pb = {__shfl_xor(B.re, 32), __shfl_xor(B.im, 32)};
            const cpx<float> a2 = hl ? pb : A, b2 = hl ? B : pa;
            A = a2;
            B = b2;
        };
        xchg(A0, B0);
        xchg(A1, B1);
        xchg(A2, B2);
        xchg(A3, B3);
#else
        asm volatile("s_nop 1\n\t"
                     "v_permlane32_swap_b32 %0, %1\n\t"
                     "v_permlane32_swap_b32 %2, %3\n\t"
                     "v_permlane32_swap_b32 %4, %5\n\t"
                     "v_permlane32_swap_b32 %6, %7\n\t"
                     "v_permlane32_swap_b32 %8, %9\n\t"
                     "v_permlane32_swap_b32 %10, %11\n\t"
                     "v_permlane32_swap_b32 %12, %13\n\t"
                     "v_permlane32_swap_b32 %14, %15\n\t"
                     "s_nop 1"
                     : "+v"(A0.re), "+v"(B0.re), "+v"(A0.im), "+v"(B0.im), "+v"(A1.re), "+v"(B1.re), "+v"(A1.im), "+v"(B1.im),
                       "+v"(A2.re), "+v"(B2.re), "+v"(A2.im), "+v"(B2.im), "+v"(A3.re), "+v"(B3.re), "+v"(A3.im), "+v"(B3.im));
#endif
    }
    const int q = base8 & 15, plo = base8 >> 4;
    const cpx<float>* t2 = twl + OFF2 + plo;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cpx<float> t[4] = {v[brev<8>(k)], v[8 + brev<8>(k)], v[brev<8>(k + 4)], v[8 + brev<8>(k + 4)]};   // p_hi = 0..3
        dft<4>(t);                                          // X[k2] in t[brev<4>(k2)]
        const int pos = q + 16 * (k + 4 * hl) + 512 * plo;
        lds[cpad(pos)] = t[0];
#pragma unroll
        for (int k2 = 1; k2 < 4; ++k2) lds[cpad(pos + 128 * k2)] = cmul(t[brev<4>(k2)], t2[(k2 - 1) * 16]);
    }
    __syncthreads();
}

// ---- the same for the 16-16-4-16 plan (H = 16384, 1024 threads, one radix-16 butterfly per thread in pass 1) ----
// Pass 2 (radix 4, stride 256) combines the pass-1 outputs k of the four butterflies b = q + 16 (p' + 16 j'), j' = 0..3. With
// butterfly (q, p') = (lane & 15, wave) and j' = lane >> 4, those sit in the lanes l, l + 16, l + 32, l + 48 of one wave: a 4 x 4
// transpose over the lane quad - v_permlane32_swap on the register pairs (k, k + 8), then v_permlane16_swap on (k, k + 4)
// and (k + 8, k + 12) - leaves in lane (hi, lo) = (j' >> 1, j' & 1) the four inputs (sources j' = 0..3 in the slots k, k + 4,
// k + 8, k + 12) of the radix-4 butterflies of outputs kk = k + 4 lo + 8 hi, k < 4. 32 swaps replace 16 ds_write_b64 +
// 16 ds_read_b64 and two barriers per transform. Measured bound (exchange skipped, results wrong): -15 % on
// k_spectral_envelope<14>, -9 % on k_spectral_envelope_long.
#define F2_SWAP8(OP, a0, b0, a1, b1, a2, b2, a3, b3, a4, b4, a5, b5, a6, b6, a7, b7)                                        \
    asm volatile("s_nop 1\n\t" OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\t" OP " %8, %9\n\t" OP      \
                 " %10, %11\n\t" OP " %12, %13\n\t" OP " %14, %15\n\ts_nop 1"                                                \
                 : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2), "+v"(a3), "+v"(b3), "+v"(a4), "+v"(b4),       \
                   "+v"(a5), "+v"(b5), "+v"(a6), "+v"(b6), "+v"(a7), "+v"(b7))
template <int NT, int PTV>
__device__ __forceinline__ void fft14_pass12_merged(cpx<float>* lds, const cpx<float>* twl, int tid, cpx<float> (&v)[PTV]) {
    constexpr int LOG2H = 14;
    static_assert(NT == 1024 && PTV >= 16 && plan_npass(LOG2H) == 4 && plan_bits(LOG2H, 0) == 4 && plan_bits(LOG2H, 1) == 4 &&
                      plan_bits(LOG2H, 2) == 2 && plan_bits(LOG2H, 3) == 4, "the 16-16-4-16 plan on 1024 threads");
    constexpr int OFF2 = plan_tw_offset(LOG2H, 2) - plan_tw_offset(LOG2H, 1);   // pass 2's table inside twl
    const int l = tid & 63, wv = tid >> 6;
    const int x = l & 15, jp = l >> 4;                  // q = x, p' = wv, pass-1 butterfly p = wv + 16 jp
    const int b = x + 16 * (wv + 16 * jp);
    const cpx<float>* src = lds + cpad(b);
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = src[j * (1024 + 64)];   // cpad(b + 1024 j)
    __syncthreads();
    dft<16>(&v[0]);                                     // X[k] in v[brev<16>(k)]
    const cpx<float>* t1 = twl + (wv + 16 * jp);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[brev<16>(k)] = cmul(v[brev<16>(k)], t1[(k - 1) * 64]);
#define F2_S(k) v[brev<16>(k)]
    // lanes (l, l + 32): slot k keeps / receives the values of output k + 8 hi from sources lo, slot k + 8 those of lo + 2
    F2_SWAP8("v_permlane32_swap_b32", F2_S(0).re, F2_S(8).re, F2_S(0).im, F2_S(8).im, F2_S(1).re, F2_S(9).re, F2_S(1).im, F2_S(9).im,
             F2_S(2).re, F2_S(10).re, F2_S(2).im, F2_S(10).im, F2_S(3).re, F2_S(11).re, F2_S(3).im, F2_S(11).im);
    F2_SWAP8("v_permlane32_swap_b32", F2_S(4).re, F2_S(12).re, F2_S(4).im, F2_S(12).im, F2_S(5).re, F2_S(13).re, F2_S(5).im,
             F2_S(13).im, F2_S(6).re, F2_S(14).re, F2_S(6).im, F2_S(14).im, F2_S(7).re, F2_S(15).re, F2_S(7).im, F2_S(15).im);
    // lanes (l, l + 16): slot k + 4 s then holds source s of output k + 4 lo + 8 hi
    F2_SWAP8("v_permlane16_swap_b32", F2_S(0).re, F2_S(4).re, F2_S(0).im, F2_S(4).im, F2_S(1).re, F2_S(5).re, F2_S(1).im, F2_S(5).im,
             F2_S(2).re, F2_S(6).re, F2_S(2).im, F2_S(6).im, F2_S(3).re, F2_S(7).re, F2_S(3).im, F2_S(7).im);
    F2_SWAP8("v_permlane16_swap_b32", F2_S(8).re, F2_S(12).re, F2_S(8).im, F2_S(12).im, F2_S(9).re, F2_S(13).re, F2_S(9).im,
             F2_S(13).im, F2_S(10).re, F2_S(14).re, F2_S(10).im, F2_S(14).im, F2_S(11).re, F2_S(15).re, F2_S(11).im, F2_S(15).im);
    const int kk0 = 4 * (jp & 1) + 8 * (jp >> 1);
    const cpx<float>* t2 = twl + OFF2 + wv;             // p' = wv
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cpx<float> t[4] = {F2_S(k), F2_S(k + 4), F2_S(k + 8), F2_S(k + 12)};   // sources j' = 0..3
        dft<4>(t);                                      // X[k2] in t[brev<4>(k2)]
        const int pos = x + 16 * (k + kk0) + 1024 * wv; // q' + 1024 p', q' = q + 16 kk
        lds[cpad(pos)] = t[0];
#pragma unroll
        for (int k2 = 1; k2 < 4; ++k2) lds[cpad(pos + 256 * k2)] = cmul(t[brev<4>(k2)], t2[(k2 - 1) * 16]);
    }
#undef F2_S
    __syncthreads();
}

// First pass of TWO transforms of the same workgroup at once (radix-16 first pass, one butterfly per thread: 4096-,
// 8192- and 16384-point rows) (the even-sample and the odd-sample transform
// of k_spectral_envelope): thread t runs butterfly t of both, so the fifteen twiddles it derives from two loaded ones
// (f2_fft_lds.h, T0REGS) serve both - 52 instructions and two loads less per row and thread than two separate first
// passes. The outputs stay in registers (v[brev<16>(k)] = output k); fft13_from_pass0 takes a transform from there.
template <int LOG2H, int NT, int PTV>
__device__ __forceinline__ void fft_pass0_pair(const cpx<float>* __restrict__ tw, int tid, cpx<float> (&va)[PTV], cpx<float> (&vb)[PTV]) {
    constexpr int R = 16, NB = (1 << LOG2H) / R;
    static_assert(NT == NB && PTV == R && plan_bits(LOG2H, 0) == 4, "one radix-16 butterfly per thread");
    dft<R>(&va[0]);
    dft<R>(&vb[0]);
    const cpx<float>* twq = tw + plan_tw_offset(LOG2H, 0) + tid;
    cpx<float> w[R];
    w[1] = twq[0];
    w[4] = twq[3 * NB];
    w[2] = cmul(w[1], w[1]);
    w[3] = cmul(w[2], w[1]);
    w[8] = cmul(w[4], w[4]);
    w[5] = cmul(w[4], w[1]);
    w[6] = cmul(w[4], w[2]);
    w[7] = cmul(w[4], w[3]);
    w[12] = cmul(w[8], w[4]);
    w[9] = cmul(w[8], w[1]);
    w[10] = cmul(w[8], w[2]);
    w[11] = cmul(w[8], w[3]);
    w[13] = cmul(w[12], w[1]);
    w[14] = cmul(w[12], w[2]);
    w[15] = cmul(w[12], w[3]);
#pragma unroll
    for (int k = 1; k < R; ++k) {
        va[brev<R>(k)] = cmul(va[brev<R>(k)], w[k]);
        vb[brev<R>(k)] = cmul(vb[brev<R>(k)], w[k]);
    }
}

// the rest of an 8192-point transform whose first pass (fft13_pass0_pair) left its outputs in v: exchange, merged passes
// 1 + 2, last pass; results in v as fft_regs_to_regs leaves them
template <int LOG2H, int PTV, int NT, bool T0REGS, int PASS = 1>
__device__ __forceinline__ void fft_remaining_passes(cpx<float>* lds, const cpx<float>* __restrict__ tw, const cpx<float>* twl, int tid,
                                                     cpx<float> (&v)[PTV]) {
    constexpr int NP = plan_npass(LOG2H);
    if constexpr (PASS < NP) {
        fft_pass<float, LOG2H, PASS, false, PASS == NP - 1, PTV, NT, false, T0REGS>(lds, tw, twl, tid, v);
        fft_remaining_passes<LOG2H, PTV, NT, T0REGS, PASS + 1>(lds, tw, twl, tid, v);
    }
}
template <int LOG2H, int PTV, int NT, bool T0REGS>
__device__ __forceinline__ void fft_from_pass0(cpx<float>* lds, const cpx<float>* __restrict__ tw, const cpx<float>* twl, int tid,
                                               cpx<float> (&v)[PTV]) {
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[cpad(tid * 16 + k)] = v[brev<16>(k)];     // pass 0 has stride 1: outputs 16 t + k
    __syncthreads();
    if constexpr (LOG2H == 13) {
        fft13_pass12_merged<NT, PTV>(lds, twl, tid, v);
        fft_pass<float, 13, 3, false, true, PTV, NT, false, T0REGS>(lds, tw, twl, tid, v);
    } else if constexpr (LOG2H == 14) {
        fft14_pass12_merged<NT, PTV>(lds, twl, tid, v);
        fft_pass<float, 14, 3, false, true, PTV, NT, false, T0REGS>(lds, tw, twl, tid, v);
    } else {
        fft_remaining_passes<LOG2H, PTV, NT, T0REGS>(lds, tw, twl, tid, v);
    }
}

// the whole 8192-point transform, registers (first-pass layout) to registers (as fft_regs_to_regs leaves them)
template <int PTV, int NT, bool T0REGS>
__device__ __forceinline__ void fft13_regs_to_regs(cpx<float>* lds, const cpx<float>* __restrict__ tw, const cpx<float>* twl, int tid,
                                                   cpx<float> (&v)[PTV]) {
#ifdef F2_KS_PLAIN_PASSES   // diagnostic: the four separate passes of f2_fft_lds.h
    fft_regs_to_regs<float, 13, PTV, NT, T0REGS>(lds, tw, twl, tid, v);
#else
    fft_pass<float, 13, 0, true, false, PTV, NT, false, T0REGS>(lds, tw, twl, tid, v);
    fft13_pass12_merged<NT, PTV>(lds, twl, tid, v);
    fft_pass<float, 13, 3, false, true, PTV, NT, false, T0REGS>(lds, tw, twl, tid, v);
#endif
}

