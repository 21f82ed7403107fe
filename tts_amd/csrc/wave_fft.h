// One-wavefront real FFT for gfx950: a frame of N = 128*E real samples (E = 8 or
// 16) lives in the registers of ONE 64-lane wave as P = N/2 = 64*E complex
// values, E per lane.  Three register-resident DFT passes (radix E, 8, 8) are
// separated by two lane<->register transposes through a wave-private LDS
// buffer; a third LDS exchange fetches the mirror bin P-k that the
// real-from-complex split needs.  No workgroup barrier is involved: a wave
// only ever reads LDS words it wrote itself.
//
// Index conventions (A = E/8):
//   time side   : complex index m = 64*n2 + lane, register n2 in [0,E),
//                 lane = 8*n1 + n0.  z[m] = x[2m] + i*x[2m+1].
//   freq side   : k = k0a + A*k0b + E*k1 + 8E*k2 with register = A*k2 + k0a and
//                 lane = 8*k0b + k1  (k0 = A*k0b + k0a is the radix-E digit).
//   "native" spectrum order: the (register, lane) order of the freq side; bin 0
//   carries the two real bins DC and Nyquist packed as (re, im).
//
// The code is plain C++ on purpose: every function is usable from a host
// emulation in which the 64 lanes are a loop (tests/csrc), which is how the
// index algebra is checked without a GPU.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GLV_HD __host__ __device__ __forceinline__
#else
#define GLV_HD inline
#endif

namespace glv {

struct cf {
    float x, y;
};
GLV_HD cf operator+(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
GLV_HD cf operator-(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
// a * b
GLV_HD cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// a * conj(b)
GLV_HD cf cmulc(cf a, cf b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }
// forward twiddles are stored as e^{-i*theta}; INV selects the conjugate
template <bool INV>
GLV_HD cf twmul(cf a, cf w) {
    return INV ? cmulc(a, w) : cmul(a, w);
}
// multiply by -i (forward) or +i (inverse)
template <bool INV>
GLV_HD cf rot90(cf a) {
    return INV ? cf{-a.y, a.x} : cf{a.y, -a.x};
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
        const cf s13 = v[1] + v[3], d13 = rot90<INV>(v[1] - v[3]);
        v[0] = s02 + s13;
        v[1] = d02 + d13;
        v[2] = s02 - s13;
        v[3] = d02 - d13;
    }
};

template <bool INV>
struct Dft<8, INV> {
    static GLV_HD void run(cf (&v)[8]) {
        constexpr float h = 0.70710678118654752440f;
        cf b[4], c[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            b[n] = v[n] + v[n + 4];
            c[n] = v[n] - v[n + 4];
        }
        // c[n] *= W8^n
        if (INV) {
            c[1] = cf{(c[1].x - c[1].y) * h, (c[1].x + c[1].y) * h};
            c[3] = cf{-(c[3].x + c[3].y) * h, (c[3].x - c[3].y) * h};
        } else {
            c[1] = cf{(c[1].x + c[1].y) * h, (c[1].y - c[1].x) * h};
            c[3] = cf{(c[3].y - c[3].x) * h, -(c[3].x + c[3].y) * h};
        }
        c[2] = rot90<INV>(c[2]);
        Dft<4, INV>::run(b);
        Dft<4, INV>::run(c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = b[k];
            v[2 * k + 1] = c[k];
        }
    }
};

template <bool INV>
struct Dft<16, INV> {
    static GLV_HD void run(cf (&v)[16]) {
        // W16^n = (cos(pi n/8), -sin(pi n/8)), n = 1..7
        constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
        constexpr float h = 0.70710678118654752440f;
        cf b[8], c[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            b[n] = v[n] + v[n + 8];
            c[n] = v[n] - v[n + 8];
        }
        c[1] = twmul<INV>(c[1], cf{c1, -s1});
        c[2] = twmul<INV>(c[2], cf{h, -h});
        c[3] = twmul<INV>(c[3], cf{s1, -c1});
        c[4] = rot90<INV>(c[4]);
        c[5] = twmul<INV>(c[5], cf{-s1, -c1});
        c[6] = twmul<INV>(c[6], cf{-h, -h});
        c[7] = twmul<INV>(c[7], cf{-c1, -s1});
        Dft<8, INV>::run(b);
        Dft<8, INV>::run(c);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[2 * k] = b[k];
            v[2 * k + 1] = c[k];
        }
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
    // transpose 1: slot = k0a*X1 + 72*k0b + 8*n1 + n0   (72 = 64 + 8 keeps the
    //   strided side on distinct banks)
    static constexpr int X1 = 7 * 72 + 64;
    // transpose 2: slot = k0a*X2 + 8*k0b + k1 + 65*n0
    static constexpr int X2 = 65 * 7 + 64 + 1;
    // mirror: slot = reg*64 + lane, plus row E holding a copy of bin 0
    static constexpr int X3 = (E + 1) * 64;
    static constexpr int XBUF = (A * X1 > X3) ? (A * X1 > A * X2 ? A * X1 : A * X2)
                                              : (X3 > A * X2 ? X3 : A * X2);
    // table sizes (cf units)
    static constexpr int TW_A = (E - 1) * 64;  // [k0-1][lane]   W_P^(k0*lane)
    static constexpr int TW_B = 7 * 8;         // [k1-1][n0]     W_64^(k1*n0)
    static constexpr int TW_P = E * 64;        // [reg][lane]    0.5 * W_N^k
    static constexpr int WIN = E * 64;         // [reg][lane]    (w[2m], w[2m+1])
    static constexpr int TABLES = TW_A + TW_B + TW_P + WIN;

    // bin index held by (reg, lane) on the frequency side
    static GLV_HD int bin(int reg, int lane) {
        return (reg % A) + A * (lane >> 3) + E * (lane & 7) + 8 * E * (reg / A);
    }
    // slot (relative to row A*(7-k2)+k0a... see mirror_read) where lane finds the
    // mirror partner of its register with low digit k0a
    static GLV_HD int mirror_base(int k0a, int lane) {
        const int low = k0a + A * (lane >> 3) + E * (lane & 7);
        if (low == 0) return A * 64;  // lane 0: partner lives in its own lane, A rows up
        const int lp = 8 * E - low;
        const int k0bp = (lp / A) % 8, k1p = lp / E;
        return 8 * k0bp + k1p;
    }
};

// Pointers to the read-only tables (LDS on the device).
struct Tables {
    const cf* twA;
    const cf* twB;
    const cf* twP;
    const cf* win;
};

// ---------------------------------------------------------------------------
// Forward complex FFT of the wave's P values: the three phases between LDS
// exchanges.  `w` = buffer this phase writes, `r` = buffer it reads (the same
// buffer on the device; the host emulation alternates two).
// ---------------------------------------------------------------------------
template <int A>
struct WaveFft {
    using G = Geo<A>;
    static constexpr int E = G::E;

    // ---- forward ----
    static GLV_HD void fwd_a(cf (&v)[E], const Tables& t, cf* w, int lane) {
        Dft<E, false>::run(v);
#pragma unroll
        for (int k0 = 1; k0 < E; ++k0) v[k0] = cmul(v[k0], t.twA[(k0 - 1) * 64 + lane]);
#pragma unroll
        for (int k0 = 0; k0 < E; ++k0) w[(k0 % A) * G::X1 + 72 * (k0 / A) + lane] = v[k0];
    }
    static GLV_HD void fwd_b(cf (&v)[E], const Tables& t, const cf* r, int lane) {
        const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) u[n1] = r[a * G::X1 + 72 * hi + 8 * n1 + lo];
            Dft<8, false>::run(u);
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) u[k1] = cmul(u[k1], t.twB[(k1 - 1) * 8 + lo]);
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) v[A * k1 + a] = u[k1];
        }
    }
    static GLV_HD void fwd_b_store(const cf (&v)[E], cf* w, int lane) {
        const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) w[a * G::X2 + 8 * hi + k1 + 65 * lo] = v[A * k1 + a];
    }
    static GLV_HD void fwd_c(cf (&v)[E], const cf* r, int lane) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#pragma unroll
            for (int n0 = 0; n0 < 8; ++n0) u[n0] = r[a * G::X2 + lane + 65 * n0];
            Dft<8, false>::run(u);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) v[A * k2 + a] = u[k2];
        }
    }

    // ---- mirror exchange: everybody publishes, then reads bin P-k ----
    static GLV_HD void mirror_store(const cf (&v)[E], cf* w, int lane) {
#pragma unroll
        for (int reg = 0; reg < E; ++reg) w[reg * 64 + lane] = v[reg];
        if (lane == 0) w[E * 64] = v[0];
    }
    // mbase[a] = Geo::mirror_base(a, lane)
    static GLV_HD void mirror_load(cf (&m)[E], const cf* r, const int (&mbase)[A]) {
#pragma unroll
        for (int reg = 0; reg < E; ++reg) {
            const int a = reg % A, k2 = reg / A;
            m[reg] = r[(A * (7 - k2) + a) * 64 + mbase[a]];
        }
    }

    // ---- real <-> half-complex split ----
    // X[k] = 0.5*(Zk + conj Zm) - i * (0.5 W_N^k) * (Zk - conj Zm); bin 0 packs
    // (DC, Nyquist) = (Re Z0 + Im Z0, Re Z0 - Im Z0).
    static GLV_HD void split_fwd(cf (&v)[E], const cf (&m)[E], const Tables& t, int lane) {
#pragma unroll
        for (int reg = 0; reg < E; ++reg) {
            const cf a = v[reg], b = m[reg];
            const cf s = {a.x + b.x, a.y - b.y};
            const cf d = {a.x - b.x, a.y + b.y};
            const cf tw = t.twP[reg * 64 + lane];
            const cf u = cmul(tw, d);
            cf x = {0.5f * s.x + u.y, 0.5f * s.y - u.x};
            if (reg == 0 && lane == 0) x = cf{a.x + a.y, a.x - a.y};
            v[reg] = x;
        }
    }
    // Z[k] = (Xk + conj Xm) + i * conj(W_N^k) * (Xk - conj Xm)   (twP holds 0.5 W,
    // so the second term is doubled); bin 0: (DC + Nyq, DC - Nyq).
    // The result is 2x the half-size spectrum; callers fold the factor into the
    // 1/N of the inverse transform.
    static GLV_HD void split_inv(cf (&v)[E], const cf (&m)[E], const Tables& t, int lane) {
#pragma unroll
        for (int reg = 0; reg < E; ++reg) {
            const cf a = v[reg], b = m[reg];
            const cf s = {a.x + b.x, a.y - b.y};
            const cf d = {a.x - b.x, a.y + b.y};
            const cf tw = t.twP[reg * 64 + lane];
            const cf u = cmulc(d, tw);  // d * conj(0.5 W)
            cf z = {s.x - 2.0f * u.y, s.y + 2.0f * u.x};
            if (reg == 0 && lane == 0) z = cf{a.x + a.y, a.x - a.y};
            v[reg] = z;
        }
    }

    // ---- inverse (unnormalised) ----
    static GLV_HD void inv_c(cf (&v)[E], cf* w, int lane) {
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) u[k2] = v[A * k2 + a];
            Dft<8, true>::run(u);
#pragma unroll
            for (int n0 = 0; n0 < 8; ++n0) w[a * G::X2 + lane + 65 * n0] = u[n0];
        }
    }
    static GLV_HD void inv_b(cf (&v)[E], const Tables& t, const cf* r, int lane) {
        const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            cf u[8];
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) u[k1] = r[a * G::X2 + 8 * hi + k1 + 65 * lo];
#pragma unroll
            for (int k1 = 1; k1 < 8; ++k1) u[k1] = cmulc(u[k1], t.twB[(k1 - 1) * 8 + lo]);
            Dft<8, true>::run(u);
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) v[A * n1 + a] = u[n1];
        }
    }
    static GLV_HD void inv_b_store(const cf (&v)[E], cf* w, int lane) {
        const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) w[a * G::X1 + 72 * hi + 8 * n1 + lo] = v[A * n1 + a];
    }
    static GLV_HD void inv_a(cf (&v)[E], const Tables& t, const cf* r, int lane) {
#pragma unroll
        for (int k0 = 0; k0 < E; ++k0) v[k0] = r[(k0 % A) * G::X1 + 72 * (k0 / A) + lane];
#pragma unroll
        for (int k0 = 1; k0 < E; ++k0) v[k0] = cmulc(v[k0], t.twA[(k0 - 1) * 64 + lane]);
        Dft<E, true>::run(v);
    }
};

}  // namespace glv
