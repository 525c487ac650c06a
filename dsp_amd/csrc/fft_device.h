// fft_device.h -- register-level building blocks of the Stockham FFT kernels (gfx950).
//
// Replaces the butterfly loop of dsp::fft0 (reference fft.h:40-54).  The reference is a
// radix-2 decimation-in-time loop with a bit-reversal copy and a twiddle recurrence; here
// every thread keeps P = 16 points in registers, a pass is one radix-16/8/4/2 butterfly per
// thread, and the autosort (Stockham) indexing makes the output come out in natural order
// with no bit-reversal pass.  SIGN = +1 is the reference's FORWARD (e^{+i 2 pi k n / N},
// fft.h:22-29), SIGN = -1 its INVERSE; neither normalises.
#pragma once

#include <hip/hip_runtime.h>

namespace dspgpu {

#define DSP_DEV __device__ __forceinline__
#define DSP_DEV_EARLY __host__ __device__ __forceinline__

// Native 2-vector, not HIP's float2 struct: its + - * are <2 x float> operations, which the backend maps 1:1
// onto v_pk_add/mul/fma_f32 with the (re, im) pair in one register pair and swaps/negations folded into
// op_sel/neg modifiers.  With the struct, the SLP vectoriser paired up unrelated scalars and a third of the
// FFT's VALU instructions were v_mov shuffles.
typedef float c32 __attribute__((ext_vector_type(2)));
DSP_DEV_EARLY c32 mk(float x, float y) { return c32{x, y}; }

// Streaming (use-once) global accesses: non-temporal loads and stores measured +3 % on a 2 GiB -> 2 GiB
// copy with this kernel's access pattern (scripts/copy_patterns2.hip).
DSP_DEV c32 load_stream(const c32 *p) {
    const float *q = reinterpret_cast<const float *>(p);
    return mk(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1));
}
DSP_DEV void store_stream(c32 *p, c32 v) {
    float *q = reinterpret_cast<float *>(p);
    __builtin_nontemporal_store(v.x, q);
    __builtin_nontemporal_store(v.y, q + 1);
}

// Workgroup -> tile map that lets each XCD stream one contiguous eighth of the batch: workgroups are
// dealt round-robin over the 8 XCDs (blocks b and b+8 share one), so tile = (b % 8) * chunk + b / 8
// keeps every XCD's L2 and its HBM channels on neighbouring addresses (+4 % on the same copy).
// Speed only; any placement gives the same result.  Launch 8 * chunk workgroups, skip tiles >= count.
DSP_DEV size_t xcd_tile(unsigned block, size_t chunk) { return static_cast<size_t>(block & 7u) * chunk + (block >> 3); }

DSP_DEV c32 cadd(c32 a, c32 b) { return a + b; }
DSP_DEV c32 csub(c32 a, c32 b) { return a - b; }
// Complex product in two packed instructions.  The swap and the negation ride on the op_sel / neg
// modifiers (lo result: a.x*b.x then -a.y*b.y + lo; hi result: a.y*b.x then a.x*b.y + hi); written as
// asm because the backend otherwise spends a v_xor and two v_mov per product on them.
// Both instructions sit in ONE asm statement: between two statements hipcc pads a dependent use of an asm result
// with `s_nop 0` (it cannot see that v_pk_mul_f32 writes whole registers), one issue slot per complex product.
DSP_DEV c32 cmul(c32 a, c32 b) {
    c32 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=&v"(r)
        : "v"(a), "v"(b));
    return r;
}

// the same with a compile-time constant b held in a scalar register pair (no VGPRs, no v_mov)
DSP_DEV c32 cmul_k(c32 a, c32 b) {
    c32 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=&v"(r)
        : "v"(a), "s"(b));
    return r;
}

// a + (SIGN i) b in one v_pk_add_f32: lo = a.x -/+ b.y, hi = a.y +/- b.x
template <int SIGN>
DSP_DEV c32 add_mul_i(c32 a, c32 b) {
    c32 r;
    if constexpr (SIGN > 0)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// multiply by SIGN * i
template <int SIGN>
DSP_DEV c32 mul_i(c32 a) {
    return add_mul_i<SIGN>(c32{0.0f, 0.0f}, a);
}

constexpr float cos16(int q) {
    constexpr float t[16] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                             0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                             -1.0f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f,
                             0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
    return t[q & 15];
}
constexpr float sin16(int q) { return cos16(q + 12); }  // sin(x) = cos(x - pi/2)

constexpr float cos64(int q) {
    constexpr float t[17] = {1.0f, 0.99518472667219693f, 0.98078528040323043f, 0.95694033573220882f,
                             0.92387953251128674f, 0.88192126434835505f, 0.83146961230254524f, 0.77301045336273699f,
                             0.70710678118654757f, 0.63439328416364549f, 0.55557023301960229f, 0.47139673682599781f,
                             0.38268343236508984f, 0.29028467725446233f, 0.19509032201612833f, 0.09801714032956077f, 0.0f};
    q &= 63;
    return q <= 16 ? t[q] : (q <= 32 ? -t[32 - q] : (q <= 48 ? -t[q - 32] : t[64 - q]));
}
constexpr float sin64(int q) { return cos64(q + 48); }  // sin(x) = cos(x - pi/2)

constexpr float cos128(int q) {
    constexpr float t[33] = {1.0f, 0.99879545620517241f, 0.99518472667219693f, 0.98917650996478101f, 0.98078528040323043f,
                             0.97003125319454397f, 0.95694033573220882f, 0.94154406518302081f, 0.92387953251128674f,
                             0.90398929312344334f, 0.88192126434835505f, 0.85772861000027212f, 0.83146961230254524f,
                             0.80320753148064494f, 0.77301045336273699f, 0.74095112535495911f, 0.70710678118654757f,
                             0.67155895484701833f, 0.63439328416364549f, 0.59569930449243347f, 0.55557023301960229f,
                             0.51410274419322166f, 0.47139673682599781f, 0.4275550934302822f, 0.38268343236508984f,
                             0.33688985339222005f, 0.29028467725446233f, 0.24298017990326398f, 0.19509032201612833f,
                             0.14673047445536175f, 0.09801714032956077f, 0.049067674327418126f, 0.0f};
    q &= 127;
    return q <= 32 ? t[q] : (q <= 64 ? -t[64 - q] : (q <= 96 ? -t[q - 64] : t[128 - q]));
}
constexpr float sin128(int q) { return cos128(q + 96); }  // sin(x) = cos(x - pi/2)

// exp(2 pi i s / P) for the register counts in use (P points per thread)
template <int P>
constexpr float cos_per_points(int s) {
    static_assert(P == 16 || P == 32 || P == 64, "points per thread");
    return P == 16 ? cos16(s) : (P == 32 ? cos64(2 * s) : cos64(s));
}
template <int P>
constexpr float sin_per_points(int s) {
    return P == 16 ? sin16(s) : (P == 32 ? sin64(2 * s) : sin64(s));
}

// multiply by exp(SIGN * 2 pi i * Q / 16), Q a compile-time constant
template <int Q, int SIGN>
DSP_DEV c32 mul_w16(c32 a) {
    constexpr int q = ((Q % 16) + 16) % 16;
    if constexpr (q == 0) return a;
    else if constexpr (q == 4) return mul_i<SIGN>(a);
    else if constexpr (q == 8) return -a;
    else if constexpr (q == 12) return mul_i<-SIGN>(a);
    else {
        constexpr float c = cos16(q);
        constexpr float s = SIGN * sin16(q);
        return cmul_k(a, c32{c, s});
    }
}

template <int SIGN>
DSP_DEV void dft2(c32 &a, c32 &b) {
    const c32 t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

// natural-order 4-point DFT with W4 = SIGN*i
template <int SIGN>
DSP_DEV void dft4(c32 &a, c32 &b, c32 &c, c32 &d) {
    const c32 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = csub(b, d);
    a = cadd(t0, t2);
    b = add_mul_i<SIGN>(t1, t3);
    c = csub(t0, t2);
    d = add_mul_i<-SIGN>(t1, t3);
}

// In-register DFT of v[0..R), natural order in and out.
template <int R, int SIGN>
struct Dft;

template <int SIGN>
struct Dft<2, SIGN> {
    static DSP_DEV void run(c32 (&v)[2]) { dft2<SIGN>(v[0], v[1]); }
};
template <int SIGN>
struct Dft<4, SIGN> {
    static DSP_DEV void run(c32 (&v)[4]) { dft4<SIGN>(v[0], v[1], v[2], v[3]); }
};
// 8 = 4 x 2: n = 2*n1 + n2, k = k1 + 4*k2
template <int SIGN>
struct Dft<8, SIGN> {
    static DSP_DEV void run(c32 (&v)[8]) {
        dft4<SIGN>(v[0], v[2], v[4], v[6]);  // n2 = 0 -> A[k1][0] at v[2*k1]
        dft4<SIGN>(v[1], v[3], v[5], v[7]);  // n2 = 1 -> A[k1][1] at v[2*k1+1]
        v[3] = mul_w16<2, SIGN>(v[3]);       // W8^{k1}
        v[5] = mul_w16<4, SIGN>(v[5]);
        v[7] = mul_w16<6, SIGN>(v[7]);
        dft2<SIGN>(v[0], v[1]);
        dft2<SIGN>(v[2], v[3]);
        dft2<SIGN>(v[4], v[5]);
        dft2<SIGN>(v[6], v[7]);
        // X[k1 + 4*k2] sits at v[2*k1 + k2]
        const c32 o1 = v[2], o2 = v[4], o3 = v[6], o4 = v[1], o5 = v[3], o6 = v[5];
        v[1] = o1;
        v[2] = o2;
        v[3] = o3;
        v[4] = o4;
        v[5] = o5;
        v[6] = o6;
    }
};
// 16 = 4 x 4: n = 4*n1 + n2, k = k1 + 4*k2
template <int SIGN>
struct Dft<16, SIGN> {
    static DSP_DEV void run(c32 (&v)[16]) {
        dft4<SIGN>(v[0], v[4], v[8], v[12]);   // column n2: A[k1][n2] lands at v[4*k1 + n2]
        dft4<SIGN>(v[1], v[5], v[9], v[13]);
        dft4<SIGN>(v[2], v[6], v[10], v[14]);
        dft4<SIGN>(v[3], v[7], v[11], v[15]);
        v[5] = mul_w16<1, SIGN>(v[5]);  // W16^{n2*k1}
        v[6] = mul_w16<2, SIGN>(v[6]);
        v[7] = mul_w16<3, SIGN>(v[7]);
        v[9] = mul_w16<2, SIGN>(v[9]);
        v[10] = mul_w16<4, SIGN>(v[10]);
        v[11] = mul_w16<6, SIGN>(v[11]);
        v[13] = mul_w16<3, SIGN>(v[13]);
        v[14] = mul_w16<6, SIGN>(v[14]);
        v[15] = mul_w16<9, SIGN>(v[15]);
        dft4<SIGN>(v[0], v[1], v[2], v[3]);  // row k1: X[k1 + 4*k2] lands at v[4*k1 + k2]
        dft4<SIGN>(v[4], v[5], v[6], v[7]);
        dft4<SIGN>(v[8], v[9], v[10], v[11]);
        dft4<SIGN>(v[12], v[13], v[14], v[15]);
        // 4x4 transpose of the register names
        c32 t;
        t = v[1]; v[1] = v[4]; v[4] = t;
        t = v[2]; v[2] = v[8]; v[8] = t;
        t = v[3]; v[3] = v[12]; v[12] = t;
        t = v[6]; v[6] = v[9]; v[9] = t;
        t = v[7]; v[7] = v[13]; v[13] = t;
        t = v[11]; v[11] = v[14]; v[14] = t;
    }
};

// multiply by exp(SIGN * 2 pi i * q / 32); q is a constant once the caller's loops are unrolled.  The constant is
// materialised in a scalar register pair at the point of use (the empty volatile asm cannot be hoisted): left to
// itself the compiler keeps every distinct constant live across a persistent loop and spills scalar registers.
template <int SIGN>
DSP_DEV c32 mul_w32(c32 a, int q) {
    q &= 31;
    if (q == 0) return a;
    if (q == 8) return mul_i<SIGN>(a);
    if (q == 16) return -a;
    if (q == 24) return mul_i<-SIGN>(a);
    c32 k = c32{cos64(2 * q), SIGN * sin64(2 * q)};
    asm volatile("" : "+s"(k));
    return cmul_k(a, k);
}

// 32 = 8 x 4: n = 4*n1 + n2, k = k1 + 8*k2.  Natural order in and out.
template <int SIGN>
struct Dft<32, SIGN> {
    static DSP_DEV void run(c32 (&v)[32]) {
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {  // A[k1][n2] = sum_n1 v[4 n1 + n2] W8^{n1 k1}, left at v[4 k1 + n2]
            c32 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = v[4 * i + n2];
            Dft<8, SIGN>::run(t);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[4 * i + n2] = t[i];
        }
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) {
#pragma unroll
            for (int n2 = 1; n2 < 4; ++n2) v[4 * k1 + n2] = mul_w32<SIGN>(v[4 * k1 + n2], n2 * k1);
        }
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) dft4<SIGN>(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
        c32 o[32];  // X[k1 + 8 k2] sits at v[4 k1 + k2]: a renaming, no instruction
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) o[k1 + 8 * k2] = v[4 * k1 + k2];
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = o[i];
    }
};

// exp(SIGN * 2 pi i * num / 2^log2den), num < 2^24 exact in fp32; sincospif is accurate
// to ~1 ulp, which keeps the transform well inside the 1e-5*log2(N) budget (the reference
// builds its twiddles by a double-precision recurrence, fft.h:43-44,52).
template <int SIGN>
DSP_DEV c32 unit_root(unsigned num, int log2den) {
    float sn, cs;
    sincospif(static_cast<float>(num) * __builtin_ldexpf(2.0f, -log2den), &sn, &cs);
    return mk(cs, SIGN > 0 ? sn : -sn);
}

// The same for exponents beyond 2^24 (the inter-step twiddle of a 2^30-point transform): the angle is an exact
// dyadic fraction in double.
template <int SIGN>
DSP_DEV c32 unit_root_wide(unsigned long long num, int log2den) {
    double sn, cs;
    sincospi(static_cast<double>(num) * __builtin_ldexp(2.0, -log2den), &sn, &cs);
    return mk(static_cast<float>(cs), static_cast<float>(SIGN > 0 ? sn : -sn));
}

// One Stockham pass on the P register-resident points of a thread.
//   N = 2^LOG2N points per transform, T = N/P threads per transform, `lt` = this thread's
//   index inside its transform.  r[m] holds element lt + T*m of the pass input.
//   The thread owns B = P/R butterflies j = lt + T*s; butterfly s reads r[s + t*B] and
//   leaves output u in r[s + u*B].  NS = product of the radices of the earlier passes.
template <int R, int NS, int LOG2N, int SIGN, int P, typename After>
DSP_DEV void stockham_pass(c32 (&r)[P], int lt, After after);

template <int R, int NS, int LOG2N, int SIGN, int P>
DSP_DEV void stockham_pass(c32 (&r)[P], int lt) {
    stockham_pass<R, NS, LOG2N, SIGN, P>(r, lt, [](int) {});
}

// `after(s)` runs once butterfly s has left its outputs in r[s + u*B]: a last pass can send them on their way while the
// next butterfly is computed
template <int R, int NS, int LOG2N, int SIGN, int P, typename After>
DSP_DEV void stockham_pass(c32 (&r)[P], int lt, After after) {
    constexpr int N = 1 << LOG2N;
    constexpr int T = N / P;
    constexpr int B = P / R;
#pragma unroll
    for (int s = 0; s < B; ++s) {
        c32 v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = r[s + t * B];
        if constexpr (NS > 1) {
            constexpr int LOG2DEN = __builtin_ctz(NS * R);
            const unsigned k = static_cast<unsigned>(lt + T * s) & (NS - 1);
            c32 pw[R];
            if constexpr (B > 1 && NS * R == N) {
                // last pass, several butterflies per thread: k_s = k_0 + T*s never wraps (lt < T, T*B = NS), and
                // T*s / N = s/P, so W_N^{k_s} = W_N^{k_0} * W_P^s with W_P^s a compile-time constant: one sincospif
                // per thread instead of one per butterfly
                const c32 w0 = unit_root<SIGN>(static_cast<unsigned>(lt) & (NS - 1), LOG2DEN);
                pw[1] = s == 0 ? w0 : cmul(w0, c32{cos_per_points<P>(s), SIGN * sin_per_points<P>(s)});
            } else {
                pw[1] = unit_root<SIGN>(k, LOG2DEN);
            }
#pragma unroll
            for (int t = 2; t < R; ++t) pw[t] = cmul(pw[t >> 1], pw[t - (t >> 1)]);
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = cmul(v[t], pw[t]);
        }
        Dft<R, SIGN>::run(v);
#pragma unroll
        for (int u = 0; u < R; ++u) r[s + u * B] = v[u];
        after(s);
    }
}

// Position inside the transform to which butterfly j sends its output u in a pass
// with radix R after sub-length NS:  (j / NS) * NS * R + (j % NS) + u * NS
template <int R, int NS>
DSP_DEV int stockham_dest(int j, int u) {
    return ((j & ~(NS - 1)) * R) + (j & (NS - 1)) + u * NS;
}

constexpr int radix_at(int log2n, int pass) {
    // greedy 16,16,...,remainder
    int left = log2n - 4 * pass;
    return left >= 4 ? 16 : (left > 0 ? (1 << left) : 1);
}

}  // namespace dspgpu
