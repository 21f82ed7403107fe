// One-wavefront real FFT for gfx950.  A frame of N = 1024*A real samples (A = 1 or 2) lives
// in the registers of ONE 64-lane wave as P = N/2 = 512*A complex values, E = 8*A per lane.
//
// The complex transform of size P is A interleaved transforms of size 512 (decimation in
// time by A), each done as three register-resident radix-8 passes separated by two
// lane<->register transposes through a wave-private LDS buffer; for A = 2 the two halves
// meet in one radix-2 butterfly that needs no data movement, because both halves hold bin
// k in the same (register, lane).  A third LDS exchange fetches the mirror bin P-k that the
// real-from-complex split needs.  No workgroup barrier is involved: a wave only ever reads
// LDS words it wrote itself.
//
// Index conventions (register q = r + 8a, r in [0,8), a in [0,A)):
//   time side   : row r, sub-sequence a, lane l hold z_a[64r + l] = (x[s], x[s+1]) with
//                 s = 128A*r + 2A*l + 2a: a lane's share of a row is 2A contiguous floats,
//                 a row is 128A contiguous samples, a hop of H samples is H/(128A) rows.
//   freq side   : lane = 8*k0 + k1, register q = k2 + 8h hold bin k = k0 + 8*k1 + 64*k2 + 512*h.
//   "native" spectrum order: the (register, lane) order of the freq side; bin 0 carries the
//   two real bins DC and Nyquist packed as (re, im).
//
// The code is plain C++ on purpose: every function is usable from a host emulation in which
// the 64 lanes are a loop (tests/csrc), which is how the index algebra is checked without a
// GPU.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GLV_HD __host__ __device__ __forceinline__
#else
#define GLV_HD inline
#endif

namespace glv {

// A complex value.  Under hipcc it is a two-float vector, so that sums and differences are
// packed-fp32 instructions on register pairs without help from the vectoriser; a host compiler
// (the emulation in tests/csrc) sees a plain struct.
#if defined(__HIP__)
typedef float cf __attribute__((ext_vector_type(2)));
#else
struct cf {
    float x, y;
};
GLV_HD cf operator+(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
GLV_HD cf operator-(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
#endif
// four floats moved as one 16-byte access
struct alignas(16) f4 {
    float x, y, z, w;
};

// Complex primitives.  On the device each is written as the packed-fp32 instructions it needs,
// with operand half selection (op_sel: which half of a source feeds the low / the high result)
// and per-half negation: a complex product is two instructions, a sum with a rotated or
// conjugated operand one.  Left to vectorise the scalar forms, the compiler computes sum and
// difference for both halves and moves one of each away - about twice the instructions.
#ifndef GLV_PK_ASM
#define GLV_PK_ASM 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && GLV_PK_ASM
#define GLV_PK2(name, text)                          \
    __device__ __forceinline__ cf name(cf a, cf b) { \
        cf r;                                        \
        asm(text : "=v"(r) : "v"(a), "v"(b));        \
        return r;                                    \
    }
// a + i*b, a - i*b, a + conj(b), a - conj(b)
GLV_PK2(addi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
GLV_PK2(subi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
GLV_PK2(add_conj, "v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1] neg_hi:[0,1]")
GLV_PK2(sub_conj, "v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1] neg_lo:[0,1]")
#undef GLV_PK2
// a * b:       t = (a.x*b.x,  a.x*b.y);  r = (-a.y*b.y + t.lo, a.y*b.x + t.hi)
__device__ __forceinline__ cf cmul(cf a, cf b) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=v"(r)
        : "v"(a), "v"(b), "v"(t));
    return r;
}
// a * conj(b): t = (a.x*b.x, -a.x*b.y);  r = ( a.y*b.y + t.lo, a.y*b.x + t.hi)
__device__ __forceinline__ cf cmulc(cf a, cf b) {
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// k*a - i*u and a + i*k*u for a real constant k (both halves of kk hold it)
__device__ __forceinline__ cf scale_subi(cf a, cf kk, cf u) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_hi:[0,0,1]"
        : "=v"(r)
        : "v"(a), "v"(kk), "v"(u));
    return r;
}
__device__ __forceinline__ cf add_scaled_i(cf a, cf kk, cf u) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=v"(r)
        : "v"(u), "v"(kk), "v"(a));
    return r;
}
// the mirror images of the two: conj(k*a) - i*conj(u) and conj(a) + i*k*conj(u)
__device__ __forceinline__ cf scale_subi_mirror(cf a, cf kk, cf u) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[1,0,1]"
        : "=v"(r)
        : "v"(a), "v"(kk), "v"(u));
    return r;
}
__device__ __forceinline__ cf add_scaled_i_mirror(cf a, cf kk, cf u) {
    cf r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[0,0,1]"
        : "=v"(r)
        : "v"(u), "v"(kk), "v"(a));
    return r;
}
#else
GLV_HD cf addi(cf a, cf b) { return cf{a.x - b.y, a.y + b.x}; }
GLV_HD cf subi(cf a, cf b) { return cf{a.x + b.y, a.y - b.x}; }
GLV_HD cf add_conj(cf a, cf b) { return cf{a.x + b.x, a.y - b.y}; }
GLV_HD cf sub_conj(cf a, cf b) { return cf{a.x - b.x, a.y + b.y}; }
GLV_HD cf cmul(cf a, cf b) { return cf{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
GLV_HD cf cmulc(cf a, cf b) { return cf{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }
GLV_HD cf scale_subi(cf a, cf kk, cf u) { return cf{kk.x * a.x + u.y, kk.y * a.y - u.x}; }
GLV_HD cf add_scaled_i(cf a, cf kk, cf u) { return cf{a.x - kk.x * u.y, a.y + kk.y * u.x}; }
GLV_HD cf scale_subi_mirror(cf a, cf kk, cf u) { return cf{kk.x * a.x - u.y, -(kk.y * a.y) - u.x}; }
GLV_HD cf add_scaled_i_mirror(cf a, cf kk, cf u) { return cf{a.x + kk.x * u.y, kk.y * u.x - a.y}; }
#endif
GLV_HD cf scale(cf a, float k) { return cf{a.x * k, a.y * k}; }
// forward twiddles are stored as e^{-i*theta}; INV selects the conjugate
template <bool INV>
GLV_HD cf twmul(cf a, cf w) {
    return INV ? cmulc(a, w) : cmul(a, w);
}
// a + w*b and a - w*b for w = -i (forward) or +i (inverse)
template <bool INV>
GLV_HD cf add_rot(cf a, cf b) {
    return INV ? addi(a, b) : subi(a, b);
}
template <bool INV>
GLV_HD cf sub_rot(cf a, cf b) {
    return INV ? subi(a, b) : addi(a, b);
}

// ---------------------------------------------------------------------------
// Register-resident DFTs, natural order in -> natural order out.
// ---------------------------------------------------------------------------
template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<4, INV> {
    static GLV_HD void run(cf (&v)[4]) {
        const cf s02 = v[0] + v[2], d02 = v[0] - v[2];
        const cf s13 = v[1] + v[3], d13 = v[1] - v[3];
        v[0] = s02 + s13;
        v[1] = add_rot<INV>(d02, d13);
        v[2] = s02 - s13;
        v[3] = sub_rot<INV>(d02, d13);
    }
};

template <bool INV>
struct Dft<8, INV> {
    static GLV_HD void run(cf (&v)[8]) {
        constexpr float h = 0.70710678118654752440;  // (no f suffix: the double-precision test build wants all digits)
        cf b[4], c[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            b[n] = v[n] + v[n + 4];
            c[n] = v[n] - v[n + 4];
        }
        Dft<4, INV>::run(b);
        // c[n] *= W8^n on the way into the second radix-4 step: W8 = h(1 -+ i), W8^2 = -+i,
        // W8^3 = -h(1 +- i)  (upper sign forward)
        const cf t1 = scale(c[1], h), t3 = scale(c[3], -h);
        const cf c1 = add_rot<INV>(t1, t1), c3 = sub_rot<INV>(t3, t3);
        const cf s02 = add_rot<INV>(c[0], c[2]), d02 = sub_rot<INV>(c[0], c[2]);
        const cf s13 = c1 + c3, d13 = c1 - c3;
        v[0] = b[0];
        v[1] = s02 + s13;
        v[2] = b[1];
        v[3] = add_rot<INV>(d02, d13);
        v[4] = b[2];
        v[5] = s02 - s13;
        v[6] = b[3];
        v[7] = sub_rot<INV>(d02, d13);
    }
};

// ---------------------------------------------------------------------------
// Geometry of the wave FFT and of its LDS exchange buffer (unit = one cf).
// ---------------------------------------------------------------------------
template <int A>
struct Geo {
    static constexpr int E = 8 * A;      // complex values per lane
    static constexpr int P = 64 * E;     // complex FFT size = n_fft / 2
    static constexpr int N = 2 * P;      // n_fft
    static constexpr int ROW = 128 * A;  // samples per row (one register of every sub-sequence)
    // transpose 1 (per sub-sequence): slot = 72*k0 + 8*n1 + n0   (72 = 64 + 8 keeps the
    //   strided side on distinct banks)
    static constexpr int X1 = 7 * 72 + 64;
    // transpose 2: slot = 8*k0 + k1 + 65*n0
    static constexpr int X2 = 65 * 7 + 64 + 1;
    // mirror (per half h): slot = k2*64 + lane, plus row 8 holding a copy of a bin 0
    static constexpr int X3 = 9 * 64;
    static constexpr int XBUF = A * X3;  // X3 >= X1 >= X2
    // table sizes (cf units).  Every table is laid out in pairs of consecutive first indices:
    // entry (k, lane) sits at pair(k, lane), so a lane's entries k and k+1 are adjacent and one
    // 16-byte LDS read fetches both.
    static constexpr int TW_A = 8 * 64;             // [k0][lane]    W_512^(k0*lane)  (k0 = 0 unused)
    static constexpr int TW_B = 8 * 8;              // [k1][n0]      W_64^(k1*n0)     (k1 = 0 unused)
    static constexpr int TW_C = (A - 1) * 8 * 64;   // [k2][lane]    W_P^k, k < 512 (A = 2)
    static constexpr int TW_P = E * 64;             // [q][lane]     0.5 * W_N^bin
    static constexpr int WIN = E * 64;              // [q][lane]     (w[s], w[s+1])
    static GLV_HD int pair(int k, int lane) { return (((k >> 1) * 64 + lane) << 1) + (k & 1); }
    static GLV_HD int pair8(int k, int n0) { return (((k >> 1) * 8 + n0) << 1) + (k & 1); }
    // entries 2*kp and 2*kp+1 of a lane with one 16-byte read
    static GLV_HD void load2(const cf* table, int kp, int lane, cf& e0, cf& e1) {
        const f4 q = reinterpret_cast<const f4*>(table)[kp * 64 + lane];
        e0 = cf{q.x, q.y};
        e1 = cf{q.z, q.w};
    }
    static GLV_HD void load2_8(const cf* table, int kp, int n0, cf& e0, cf& e1) {
        const f4 q = reinterpret_cast<const f4*>(table)[kp * 8 + n0];
        e0 = cf{q.x, q.y};
        e1 = cf{q.z, q.w};
    }
    // a lane's 8 entries of a [k][lane] table (k0 .. k0+7, k0 even)
    static GLV_HD void load8(const cf* table, int k0, int lane, cf (&w)[8]) {
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) load2(table, k0 / 2 + kp, lane, w[2 * kp], w[2 * kp + 1]);
    }
    static constexpr int TABLES = TW_A + TW_B + TW_C + TW_P + WIN;

    // bin held by (register, lane) on the frequency side
    static GLV_HD int bin(int q, int lane) {
        return (lane >> 3) + 8 * (lane & 7) + 64 * (q & 7) + 512 * (q >> 3);
    }
    // inverse of bin(): where bin k < P lives on the frequency side
    static GLV_HD void locate(int k, int& q, int& lane) {
        lane = 8 * (k & 7) + ((k >> 3) & 7);
        q = ((k >> 6) & 7) + 8 * (k >> 9);
    }
    // first sample (of two) held by (register, lane) on the time side, relative to the frame
    static GLV_HD int sample(int q, int lane) { return ROW * (q & 7) + 2 * A * lane + 2 * (q >> 3); }
    // lane offset (slots) at which a lane finds the mirror partners of its bins inside the
    // other half's mirror region: partner of local bin kl is 512 - kl, i.e. row 7-k2 at this
    // offset - except for lane 0, whose partners sit in its own column one row further down
    static GLV_HD int mirror_base(int lane) {
        const int low = (lane >> 3) + 8 * (lane & 7);
        if (low == 0) return 64;
        const int lp = 64 - low;
        return 8 * (lp & 7) + (lp >> 3);
    }
};

// Pointers to the read-only tables (LDS on the device).
struct Tables {
    const cf* twA;
    const cf* twB;
    const cf* twC;
    const cf* twP;
    const cf* win;
};

// ---------------------------------------------------------------------------
// The phases of the transform between LDS exchanges.  `w` = buffer this phase writes, `r` =
// buffer it reads (the same buffer on the device; the host emulation alternates two).
// ---------------------------------------------------------------------------
// The transposes between the first and the second pass swap a lane's register index with the
// high digit (bits 3..5) of its lane number and leave the low digit alone.  GLV_XCHG_REGS = 1
// does that without LDS: lane bit 5 with v_permlane32_swap, bit 4 with v_permlane16_swap, bit 3
// with a row rotation by 8 under a bank mask (device code only; the host emulation keeps the
// LDS form, which defines the result).
#ifndef GLV_XCHG_REGS
#define GLV_XCHG_REGS 1
#endif
#if GLV_XCHG_REGS && defined(__HIP_DEVICE_COMPILE__)
#define GLV_XCHG_REGS_DEVICE 1
typedef unsigned glv_u2 __attribute__((ext_vector_type(2)));
// (x: register with the index bit clear, y: with it set) <-> lane bit 5 / 4 / 3
__device__ __forceinline__ void swap_bit5(cf& x, cf& y) {
    const glv_u2 r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.x), __float_as_uint(y.x), false, false);
    const glv_u2 r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.y), __float_as_uint(y.y), false, false);
    x = cf{__uint_as_float(r0[0]), __uint_as_float(r1[0])};
    y = cf{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
}
__device__ __forceinline__ void swap_bit4(cf& x, cf& y) {
    const glv_u2 r0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x.x), __float_as_uint(y.x), false, false);
    const glv_u2 r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x.y), __float_as_uint(y.y), false, false);
    x = cf{__uint_as_float(r0[0]), __uint_as_float(r1[0])};
    y = cf{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
}
// row_ror:8 = lane ^ 8 inside a row of 16; banks 2,3 are the lanes with bit 3 set
__device__ __forceinline__ float take_bit3_set(float keep, float from) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(from), 0x128, 0xF, 0xC, false));
}
__device__ __forceinline__ float take_bit3_clear(float keep, float from) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep), __float_as_int(from), 0x128, 0xF, 0x3, false));
}
__device__ __forceinline__ void swap_bit3(cf& x, cf& y) {
    const cf xn = cf{take_bit3_set(x.x, y.x), take_bit3_set(x.y, y.y)};
    const cf yn = cf{take_bit3_clear(y.x, x.x), take_bit3_clear(y.y, x.y)};
    x = xn;
    y = yn;
}
__device__ __forceinline__ void transpose_hi(cf (&u)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) swap_bit5(u[q], u[q + 4]);
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (!(q & 2)) swap_bit4(u[q], u[q + 2]);
#pragma unroll
    for (int q = 0; q < 8; q += 2) swap_bit3(u[q], u[q + 1]);
}
#else
#define GLV_XCHG_REGS_DEVICE 0
#endif

template <int A>
struct WaveFft {
    using G = Geo<A>;
    static constexpr int E = G::E;

    // (twr: the lane's 8 twiddles of the pass where the caller keeps them in registers, else null)
    // ---- forward: A interleaved 512-point transforms ----
    static GLV_HD void fwd_a(cf (&v)[E], const Tables& t, cf* w, int lane, const cf* twr = nullptr) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) u[n2] = v[n2 + 8 * a];
            Dft<8, false>::run(u);
            cf tw[8];
            if (twr) {
#pragma unroll
                for (int k = 0; k < 8; ++k) tw[k] = twr[k];
            } else {
                G::load8(t.twA, 0, lane, tw);
            }
#pragma unroll
            for (int k0 = 1; k0 < 8; ++k0) u[k0] = cmul(u[k0], tw[k0]);
#if GLV_XCHG_REGS_DEVICE
            transpose_hi(u);
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) v[n1 + 8 * a] = u[n1];
#else
#pragma unroll
            for (int k0 = 0; k0 < 8; ++k0) w[a * G::X1 + 72 * k0 + lane] = u[k0];
#endif
        }
    }
    static GLV_HD void fwd_b(cf (&v)[E], const Tables& t, const cf* r, int lane, const cf* twr = nullptr) {
        const int hi = lane >> 3, lo = lane & 7;
        (void)hi;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#if GLV_XCHG_REGS_DEVICE
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) u[n1] = v[n1 + 8 * a];
#else
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) u[n1] = r[a * G::X1 + 72 * hi + 8 * n1 + lo];
#endif
            Dft<8, false>::run(u);
            cf tw[8];
            if (twr) {
#pragma unroll
                for (int k = 0; k < 8; ++k) tw[k] = twr[k];
            } else {
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) G::load2_8(t.twB, kp, lo, tw[2 * kp], tw[2 * kp + 1]);
            }
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) u[k1] = cmul(u[k1], tw[k1]);
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) v[k1 + 8 * a] = u[k1];
        }
    }
    static GLV_HD void fwd_b_store(const cf (&v)[E], cf* w, int lane) {
        const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) w[a * G::X2 + 8 * hi + k1 + 65 * lo] = v[k1 + 8 * a];
    }
    // last pass, then (A = 2) the radix-2 butterfly that joins the two sub-sequences:
    // Z[k] = E[k] + W_P^k O[k], Z[k + 512] = E[k] - W_P^k O[k]
    static GLV_HD void fwd_c(cf (&v)[E], const Tables& t, const cf* r, int lane) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#pragma unroll
            for (int n0 = 0; n0 < 8; ++n0) u[n0] = r[a * G::X2 + lane + 65 * n0];
            Dft<8, false>::run(u);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) v[k2 + 8 * a] = u[k2];
        }
        if (A == 2) {
            cf tw[8];
            G::load8(t.twC, 0, lane, tw);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const cf e = v[k2], o = cmul(v[k2 + 8 * (A - 1)], tw[k2]);
                v[k2] = e + o;
                v[k2 + 8 * (A - 1)] = e - o;
            }
        }
    }

    // ---- mirror exchange: everybody publishes, then reads bin P-k ----
    // Half h (registers 8h..8h+7) is published in region h; the partner of a bin of half h
    // lives in half A-1-h.  Row 8 of a region holds the bin 0 the OTHER region's lane 0 needs.
    static GLV_HD void mirror_store(const cf (&v)[E], cf* w, int lane) {
#pragma unroll
        for (int q = 0; q < E; ++q) w[(q >> 3) * G::X3 + (q & 7) * 64 + lane] = v[q];
        if (lane == 0) {
#pragma unroll
            for (int h = 0; h < A; ++h) w[(A - 1 - h) * G::X3 + 8 * 64] = v[8 * h];
        }
    }
    // mbase = Geo::mirror_base(lane).  The *_part forms handle registers q0 .. q0+NQ-1 only
    // (m[i] pairs with v[q0 + i]), so a caller can bound how many partners are live at once.
    template <int NQ>
    static GLV_HD void mirror_load_part(cf (&m)[NQ], const cf* r, int mbase, int q0) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int h = (q0 + i) >> 3, k2 = (q0 + i) & 7;
            m[i] = r[(A - 1 - h) * G::X3 + (7 - k2) * 64 + mbase];
        }
    }
    static GLV_HD void mirror_load(cf (&m)[E], const cf* r, int mbase) { mirror_load_part<E>(m, r, mbase, 0); }

    // ---- real <-> half-complex split ----
    // X[k] = 0.5*(Zk + conj Zm) - i * (0.5 W_N^k) * (Zk - conj Zm); bin 0 packs
    // (DC, Nyquist) = (Re Z0 + Im Z0, Re Z0 - Im Z0).
    // (twp: this lane's E entries of the twP table if the caller keeps them in registers, else null)
    static GLV_HD void split_fwd(cf (&v)[E], const cf (&m)[E], const Tables& t, int lane, const cf* twp = nullptr) {
        split_fwd_part<E>(v, m, t, lane, 0, twp);
    }
    template <int NQ>
    static GLV_HD void split_fwd_part(cf (&v)[E], const cf (&m)[NQ], const Tables& t, int lane, int q0,
                                      const cf* twp = nullptr) {
        const cf half = cf{0.5f, 0.5f};
        static_assert(NQ % 2 == 0, "table entries are fetched in pairs");
        cf tw[NQ];
        if (twp) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) tw[i] = twp[q0 + i];
        } else {
#pragma unroll
            for (int kp = 0; kp < NQ / 2; ++kp) G::load2(t.twP, q0 / 2 + kp, lane, tw[2 * kp], tw[2 * kp + 1]);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = q0 + i;
            const cf a = v[q], b = m[i];
            const cf s = add_conj(a, b), d = sub_conj(a, b);
            const cf u = cmul(tw[i], d);
            cf x = scale_subi(s, half, u);  // 0.5*s - i*u
            if (q == 0 && lane == 0) x = cf{a.x + a.y, a.x - a.y};
            v[q] = x;
        }
    }
    // Z[k] = (Xk + conj Xm) + i * conj(W_N^k) * (Xk - conj Xm)   (twP holds 0.5 W,
    // so the second term is doubled); bin 0: (DC + Nyq, DC - Nyq).
    // The result is 2x the half-size spectrum; callers fold the factor into the
    // 1/N of the inverse transform.
    static GLV_HD void split_inv(cf (&v)[E], const cf (&m)[E], const Tables& t, int lane, const cf* twp = nullptr) {
        split_inv_part<E>(v, m, t, lane, 0, twp);
    }
    template <int NQ>
    static GLV_HD void split_inv_part(cf (&v)[E], const cf (&m)[NQ], const Tables& t, int lane, int q0,
                                      const cf* twp = nullptr) {
        const cf two = cf{2.0f, 2.0f};
        static_assert(NQ % 2 == 0, "table entries are fetched in pairs");
        cf tw[NQ];
        if (twp) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) tw[i] = twp[q0 + i];
        } else {
#pragma unroll
            for (int kp = 0; kp < NQ / 2; ++kp) G::load2(t.twP, q0 / 2 + kp, lane, tw[2 * kp], tw[2 * kp + 1]);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = q0 + i;
            const cf a = v[q], b = m[i];
            const cf s = add_conj(a, b), d = sub_conj(a, b);
            const cf u = cmulc(d, tw[i]);                 // d * conj(0.5 W)
            cf z = add_scaled_i(s, two, u);               // s + 2i*u
            if (q == 0 && lane == 0) z = cf{a.x + a.y, a.x - a.y};
            v[q] = z;
        }
    }

    // ---- the split done on mirror PAIRS ----
    // A bin k and its mirror P-k go through the same sums: with s = Zk + conj Zm, d = Zk - conj Zm,
    // u = (0.5 W_N^k) d the two half-complex bins are X[k] = 0.5 s - i u and X[m] = conj(0.5 s) -
    // i conj(u) (W_N^(P-k) = -conj W_N^k), and likewise on the way back.  So one lane of each
    // mirror pair of lanes can do both bins: a lane keeps the LOW half of its registers (pair i =
    // register i, bin k) and is sent the partner lane's register E-1-i (bin P-k), works on both,
    // and sends the partner's result back.  Per frame that is one full exchange instead of two,
    // six instead of ten operations per pair and direction, and half the twiddles.
    // Lane 0 is its own partner (its mirrors are E-i, one row further down - the same offset
    // trick as mirror_base); its bin 0 packs (DC, Nyquist) and its register E/2 holds the
    // self-mirrored bin P/2, which the caller treats on its own.
    static constexpr int NP = E / 2;  // pairs per lane
    // exchange 1: every lane publishes its high registers
    static GLV_HD void pair_store(const cf (&v)[E], cf* w, int lane) {
#pragma unroll
        for (int q = NP; q < E; ++q) w[(q - NP) * 64 + lane] = v[q];
    }
    // partners of pairs i0 .. i0+NQ-1 (register E-1-i of the partner lane)
    template <int NQ>
    static GLV_HD void pair_partners(cf (&m)[NQ], const cf* r, int mbase, int i0) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) m[j] = r[(NP - 1 - (i0 + j)) * 64 + mbase];
    }
    // exchange 2: the partner's finished register goes back to where it will read it
    static GLV_HD void pair_send(cf zm, cf* w, int mbase, int i) { w[(NP - 1 - i) * 64 + mbase] = zm; }
    static GLV_HD void pair_receive(cf (&v)[E], const cf* r, int lane) {
#pragma unroll
        for (int q = NP; q < E; ++q) v[q] = r[(q - NP) * 64 + lane];
    }
    // forward: (a, b) = (Z[k], Z[P-k]) -> (X[k], X[P-k]); tw = 0.5 W_N^k
    static GLV_HD void pair_fwd(cf a, cf b, cf tw, cf& xk, cf& xm) {
        const cf half = cf{0.5f, 0.5f};
        const cf s = add_conj(a, b), d = sub_conj(a, b);
        const cf u = cmul(tw, d);
        xk = scale_subi(s, half, u);
        xm = scale_subi_mirror(s, half, u);
    }
    static GLV_HD cf pair_fwd_low(cf a, cf b, cf tw) {
        const cf half = cf{0.5f, 0.5f};
        const cf s = add_conj(a, b), d = sub_conj(a, b);
        return scale_subi(s, half, cmul(tw, d));
    }
    // inverse: (a, b) = (X[k], X[P-k]) -> 2 (Z[k], Z[P-k])
    static GLV_HD void pair_inv(cf a, cf b, cf tw, cf& zk, cf& zm) {
        const cf two = cf{2.0f, 2.0f};
        const cf s = add_conj(a, b), d = sub_conj(a, b);
        const cf u = cmulc(d, tw);
        zk = add_scaled_i(s, two, u);
        zm = add_scaled_i_mirror(s, two, u);
    }
    // the same with X[P-k] = 0 (a bin whose target magnitude is zero)
    static GLV_HD void pair_inv_low_only(cf a, cf tw, cf& zk, cf& zm) {
        const cf two = cf{2.0f, 2.0f};
        const cf u = cmulc(a, tw);
        zk = add_scaled_i(a, two, u);
        zm = add_scaled_i_mirror(a, two, u);
    }

    // ---- inverse (unnormalised) ----
    // (A = 2) undo the joining butterfly: E = Z[k] + Z[k+512], O = (Z[k] - Z[k+512]) conj(W_P^k)
    static GLV_HD void inv_c(cf (&v)[E], const Tables& t, cf* w, int lane) {
        if (A == 2) {
            cf tw[8];
            G::load8(t.twC, 0, lane, tw);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const cf z0 = v[k2], z1 = v[k2 + 8 * (A - 1)];
                v[k2] = z0 + z1;
                v[k2 + 8 * (A - 1)] = cmulc(z0 - z1, tw[k2]);
            }
        }
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) u[k2] = v[k2 + 8 * a];
            Dft<8, true>::run(u);
#pragma unroll
            for (int n0 = 0; n0 < 8; ++n0) w[a * G::X2 + lane + 65 * n0] = u[n0];
        }
    }
    static GLV_HD void inv_b(cf (&v)[E], const Tables& t, const cf* r, int lane, const cf* twr = nullptr) {
        const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) u[k1] = r[a * G::X2 + 8 * hi + k1 + 65 * lo];
            cf tw[8];
            if (twr) {
#pragma unroll
                for (int k = 0; k < 8; ++k) tw[k] = twr[k];
            } else {
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) G::load2_8(t.twB, kp, lo, tw[2 * kp], tw[2 * kp + 1]);
            }
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) u[k1] = cmulc(u[k1], tw[k1]);
            Dft<8, true>::run(u);
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) v[n1 + 8 * a] = u[n1];
        }
    }
    static GLV_HD void inv_b_store(cf (&v)[E], cf* w, int lane) {
#if GLV_XCHG_REGS_DEVICE
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) u[n1] = v[n1 + 8 * a];
            transpose_hi(u);
#pragma unroll
            for (int k0 = 0; k0 < 8; ++k0) v[k0 + 8 * a] = u[k0];
        }
#else
        const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) w[a * G::X1 + 72 * hi + 8 * n1 + lo] = v[n1 + 8 * a];
#endif
    }
    static GLV_HD void inv_a(cf (&v)[E], const Tables& t, const cf* r, int lane, const cf* twr = nullptr) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#if GLV_XCHG_REGS_DEVICE
#pragma unroll
            for (int k0 = 0; k0 < 8; ++k0) u[k0] = v[k0 + 8 * a];
#else
#pragma unroll
            for (int k0 = 0; k0 < 8; ++k0) u[k0] = r[a * G::X1 + 72 * k0 + lane];
#endif
            cf tw[8];
            if (twr) {
#pragma unroll
                for (int k = 0; k < 8; ++k) tw[k] = twr[k];
            } else {
                G::load8(t.twA, 0, lane, tw);
            }
#pragma unroll
            for (int k0 = 1; k0 < 8; ++k0) u[k0] = cmulc(u[k0], tw[k0]);
            Dft<8, true>::run(u);
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) v[n2 + 8 * a] = u[n2];
        }
    }
};

}  // namespace glv
