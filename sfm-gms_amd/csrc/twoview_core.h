// twoview_core.h -- the arithmetic of the two-view stage (SfMUtil.cpp:39-82), one correspondence sample or one point at a time, in
// fp64: Nister's five-point minimal solver for cv::findEssentialMat's RANSAC, cv::RNG, RANSACUpdateNumIters, the Sampson-style
// error of EMEstimatorCallback, cv::decomposeEssentialMat, the DLT of cv::triangulatePoints, cv::undistortPoints.
//
// Everything here is a plain function of its arguments (no thread indices, no memory spaces) so that the SAME source is what the
// kernels of twoview_kernels.hip run per lane and what tests/cpp/twoview_host.cpp compiles with g++ to check the solver against
// the tests' numpy restatement on the CPU -- a test build; the product library only ever runs it on the GPU.
// Arrays a lane indexes with run-time subscripts live behind an accessor type `Mem` (double& operator()(int)): LDS words strided by
// the number of solver lanes on the device, a plain array on the host.
//
// OpenCV 4.5.2's calib3d is vendored by the reference as an import library only: parity unpinned; the algorithm is restated from
// its published form (Nister 2004; five-point.cpp / ptsetreg.cpp structure). Two things an SVD leaves open are fixed by definition
// (include/gms.h): every E carries the sign that makes its largest-magnitude entry positive, and the models of one sample are taken
// in ascending order of E[0], E[1], ...
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GMS_HD __host__ __device__ __forceinline__
#define GMS_UNROLL _Pragma("unroll")
#else
#define GMS_HD inline
#define GMS_UNROLL
#endif

namespace gms {
namespace tv {

// ---- cv::RNG (multiply-with-carry), cv::RANSACUpdateNumIters ------------------------------------------------------------------------
struct CvRng {
    uint64_t state;
    GMS_HD void seed(uint64_t s) { state = s ? s : 0xFFFFFFFFull; }
    GMS_HD uint32_t next()
    {
        state = (uint64_t)(uint32_t)state * 4164903690ull + (state >> 32);
        return (uint32_t)state;
    }
    GMS_HD int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a)) + a; }
    // RANSACPointSetRegistrator::getSubset for modelPoints = 5: each index drawn until it differs from the ones before it
    GMS_HD void sample5(int count, int idx[5])
    {
        for (int i = 0; i < 5; ++i) {
            int v = uniform(0, count);
            for (;;) {
                bool dup = false;
                for (int j = 0; j < i; ++j) dup |= idx[j] == v;
                if (!dup) break;
                v = uniform(0, count);
            }
            idx[i] = v;
        }
    }
};

GMS_HD int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = fmin(fmax(p, 0.0), 1.0);
    ep = fmin(fmax(ep, 0.0), 1.0);
    double num = fmax(1.0 - p, 2.2250738585072014e-308);
    double denom = 1.0 - pow(1.0 - ep, (double)model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    num = log(num);
    denom = log(denom);
    return (denom >= 0.0 || -num >= max_iters * (-denom)) ? max_iters : (int)rint(num / denom);
}

// ---- EMEstimatorCallback::computeError: (x2^T E x1)^2 / ((E x1)_0^2 + (E x1)_1^2 + (E^T x2)_0^2 + (E^T x2)_1^2), as fp32 ---------------
GMS_HD float sampson_error(const double* E, double x1, double y1, double x2, double y2)
{
    const double a0 = E[0] * x1 + E[1] * y1 + E[2], a1 = E[3] * x1 + E[4] * y1 + E[5], a2 = E[6] * x1 + E[7] * y1 + E[8];  // E x1
    const double b0 = E[0] * x2 + E[3] * y2 + E[6], b1 = E[1] * x2 + E[4] * y2 + E[7];                                    // E^T x2
    const double num = x2 * a0 + y2 * a1 + a2;
    return (float)(num * num / (a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1));
}

// ---- polynomials in (x, y, z) of degree <= 3 --------------------------------------------------------------------------------------------
// linear:    [x, y, z, 1]
// quadratic: [x^2, y^2, z^2, xy, xz, yz, x, y, z, 1]
// cubic:     the elimination order -- [x^3, y^3, x^2 y, x y^2, x^2 z, x^2, y^2 z, y^2, xyz, xy | x z^2, xz, x, y z^2, yz, y, z^3, z^2, z, 1]
GMS_HD constexpr int lin_lin(int a, int b)
{   // index of the product of two linear monomials among the quadratic ones
    const int t[4][4] = {{0, 3, 4, 6}, {3, 1, 5, 7}, {4, 5, 2, 8}, {6, 7, 8, 9}};
    return t[a][b];
}
GMS_HD constexpr int quad_lin(int q, int l)
{   // index of (quadratic monomial q) * (linear monomial l) among the cubic ones
    //                 x   y   z   1
    const int t[10][4] = {{0, 2, 4, 5},      // x^2
                          {3, 1, 6, 7},      // y^2
                          {10, 13, 16, 17},  // z^2
                          {2, 3, 8, 9},      // xy
                          {4, 8, 10, 11},    // xz
                          {8, 6, 13, 14},    // yz
                          {5, 9, 11, 12},    // x
                          {9, 7, 14, 15},    // y
                          {11, 14, 17, 18},  // z
                          {12, 15, 18, 19}}; // 1
    return t[q][l];
}

template <class Mem> struct FivePointMem {
    // what one solve keeps in run-time indexed memory: the constraint matrix, the null-space basis, the reduced 3 x 3 system
    Mem A;      // [10][20] constraint matrix (row-major), reduced in place
    Mem basis;  // [4][9]  X, Y, Z, W (row-major 3 x 3 each)
    Mem work;   // [60]    E E^T (6 quadratics) while A is built; afterwards B (39), the determinant (11) and the roots
};

// Null space of the 5 x 9 system Q e = 0 (rows [x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1]: Q e = x2^T E x1 for e = vec(E) row-major):
// Gauss-Jordan with full pivoting, then the four null vectors are orthonormalised (modified Gram-Schmidt, twice) -- any orthonormal
// basis of the null space yields the same essential matrices. Returns false when the system has rank below five.
template <class Mem>
GMS_HD bool null_space_5x9(const double x1[5], const double y1[5], const double x2[5], const double y2[5], Mem& basis, Mem& q /* [45] scratch */)
{
    for (int r = 0; r < 5; ++r) {
        q(r * 9 + 0) = x2[r] * x1[r];
        q(r * 9 + 1) = x2[r] * y1[r];
        q(r * 9 + 2) = x2[r];
        q(r * 9 + 3) = y2[r] * x1[r];
        q(r * 9 + 4) = y2[r] * y1[r];
        q(r * 9 + 5) = y2[r];
        q(r * 9 + 6) = x1[r];
        q(r * 9 + 7) = y1[r];
        q(r * 9 + 8) = 1.0;
    }
    int colperm[9];
    for (int c = 0; c < 9; ++c) colperm[c] = c;
    double scale = 0.0;
    for (int i = 0; i < 45; ++i) scale = fmax(scale, fabs(q(i)));
    for (int k = 0; k < 5; ++k) {
        int pr = k, pc = k;
        double best = -1.0;
        for (int r = k; r < 5; ++r)
            for (int c = k; c < 9; ++c) {
                const double v = fabs(q(r * 9 + c));
                if (v > best) {
                    best = v;
                    pr = r;
                    pc = c;
                }
            }
        if (!(best > 1e-13 * scale)) return false;
        for (int c = 0; c < 9; ++c) {  // row swap
            const double t = q(k * 9 + c);
            q(k * 9 + c) = q(pr * 9 + c);
            q(pr * 9 + c) = t;
        }
        for (int r = 0; r < 5; ++r) {  // column swap
            const double t = q(r * 9 + k);
            q(r * 9 + k) = q(r * 9 + pc);
            q(r * 9 + pc) = t;
        }
        {
            const int t = colperm[k];
            colperm[k] = colperm[pc];
            colperm[pc] = t;
        }
        const double inv = 1.0 / q(k * 9 + k);
        for (int c = k; c < 9; ++c) q(k * 9 + c) *= inv;
        for (int r = 0; r < 5; ++r) {
            if (r == k) continue;
            const double f = q(r * 9 + k);
            if (f == 0.0) continue;
            for (int c = k; c < 9; ++c) q(r * 9 + c) -= f * q(k * 9 + c);
        }
    }
    // reduced form [I | F]: null vector b has e[colperm[5 + b]] = 1, e[colperm[r]] = -F[r][b]
    for (int b = 0; b < 4; ++b) {
        for (int c = 0; c < 9; ++c) basis(b * 9 + c) = 0.0;
        basis(b * 9 + colperm[5 + b]) = 1.0;
        for (int r = 0; r < 5; ++r) basis(b * 9 + colperm[r]) = -q(r * 9 + 5 + b);
    }
    for (int pass = 0; pass < 2; ++pass)
        for (int b = 0; b < 4; ++b) {
            for (int a = 0; a < b; ++a) {
                double d = 0.0;
                for (int c = 0; c < 9; ++c) d += basis(a * 9 + c) * basis(b * 9 + c);
                for (int c = 0; c < 9; ++c) basis(b * 9 + c) -= d * basis(a * 9 + c);
            }
            double n2 = 0.0;
            for (int c = 0; c < 9; ++c) n2 += basis(b * 9 + c) * basis(b * 9 + c);
            const double inv = 1.0 / sqrt(n2);
            for (int c = 0; c < 9; ++c) basis(b * 9 + c) *= inv;
        }
    return true;
}

// The ten cubic constraints on E = x X + y Y + z Z + W -- det E = 0 and the nine entries of 2 E E^T E - tr(E E^T) E = 0 -- as the rows
// of A (10 x 20, cubic monomial order above).
template <class Mem>
GMS_HD void constraint_matrix(const Mem& basis, Mem& A, Mem& eet /* [60] */)
{
    // E[r][c] as a linear polynomial: coefficient of monomial b is basis(b * 9 + 3 r + c)
    // E E^T: six quadratics (r <= c), eet(idx6(r, c) * 10 + q)
    auto idx6 = [](int r, int c) -> int { return r <= c ? (r == 0 ? c : (r == 1 ? 2 + c : 5)) : (c == 0 ? r : (c == 1 ? 2 + r : 5)); };
    for (int i = 0; i < 60; ++i) eet(i) = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = r; c < 3; ++c) {
            const int o = idx6(r, c) * 10;
            for (int k = 0; k < 3; ++k)
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) eet(o + lin_lin(a, b)) += basis(a * 9 + 3 * r + k) * basis(b * 9 + 3 * c + k);
        }
    for (int i = 0; i < 200; ++i) A(i) = 0.0;
    // row 0: det E = E00 (E11 E22 - E12 E21) - E01 (E10 E22 - E12 E20) + E02 (E10 E21 - E11 E20)
    {
        const int minors[3][5] = {{0, 4, 8, 5, 7}, {1, 3, 8, 5, 6}, {2, 3, 7, 4, 6}};  // entry, then (p q - r s) as flat indices of E
        for (int t = 0; t < 3; ++t) {
            double quad[10];
GMS_UNROLL
            for (int i = 0; i < 10; ++i) quad[i] = 0.0;
GMS_UNROLL
            for (int a = 0; a < 4; ++a)
GMS_UNROLL
                for (int b = 0; b < 4; ++b) {
                    const double v = basis(a * 9 + minors[t][1]) * basis(b * 9 + minors[t][2]) - basis(a * 9 + minors[t][3]) * basis(b * 9 + minors[t][4]);
                    // lin_lin with run-time a, b: the table is symmetric and tiny
                    const int qi = lin_lin(a, b);
                    quad[qi] += v;
                }
            const double sg = t == 1 ? -1.0 : 1.0;
GMS_UNROLL
            for (int qi = 0; qi < 10; ++qi)
GMS_UNROLL
                for (int l = 0; l < 4; ++l) A(quad_lin(qi, l)) += sg * quad[qi] * basis(l * 9 + minors[t][0]);
        }
    }
    // rows 1..9: (2 E E^T E - tr(E E^T) E)[r][c]
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const int row = (1 + 3 * r + c) * 20;
            for (int k = 0; k < 3; ++k) {
                const int o = idx6(r, k) * 10;
                for (int qi = 0; qi < 10; ++qi) {
                    const double v = 2.0 * eet(o + qi);
                    for (int l = 0; l < 4; ++l) A(row + quad_lin(qi, l)) += v * basis(l * 9 + 3 * k + c);
                }
            }
            for (int qi = 0; qi < 10; ++qi) {
                const double tr = eet(0 * 10 + qi) + eet(3 * 10 + qi) + eet(5 * 10 + qi);  // (0,0), (1,1), (2,2)
                for (int l = 0; l < 4; ++l) A(row + quad_lin(qi, l)) -= tr * basis(l * 9 + 3 * r + c);
            }
        }
}

// A <- A[:, :10]^-1 A: Gauss-Jordan with partial pivoting on the first ten columns. false: singular.
template <class Mem>
GMS_HD bool eliminate_10x20(Mem& A)
{
    double scale = 0.0;
    for (int i = 0; i < 200; ++i) scale = fmax(scale, fabs(A(i)));
    for (int k = 0; k < 10; ++k) {
        int pr = k;
        double best = fabs(A(k * 20 + k));
        for (int r = k + 1; r < 10; ++r) {
            const double v = fabs(A(r * 20 + k));
            if (v > best) {
                best = v;
                pr = r;
            }
        }
        if (!(best > 1e-14 * scale)) return false;
        if (pr != k)
            for (int c = k; c < 20; ++c) {
                const double t = A(k * 20 + c);
                A(k * 20 + c) = A(pr * 20 + c);
                A(pr * 20 + c) = t;
            }
        const double inv = 1.0 / A(k * 20 + k);
        for (int c = k; c < 20; ++c) A(k * 20 + c) *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == k) continue;
            const double f = A(r * 20 + k);
            if (f == 0.0) continue;
            for (int c = k; c < 20; ++c) A(r * 20 + c) -= f * A(k * 20 + c);
        }
    }
    return true;
}

// Rows 4..9 of the reduced system read  x^2 z, x^2, y^2 z, y^2, xyz, xy  = -(polynomial in the last ten monomials). Row (4 + 2 i) minus
// z times row (5 + 2 i) is free of the leading monomial: B(z) [x, y, 1]^T = 0 with B[i] = (cubic, cubic, quartic) in z, coefficients
// highest power first: B(i * 13 + 0..3) for x, 4..7 for y, 8..12 for 1.
template <class Mem>
GMS_HD void reduced_system(const Mem& A, Mem& B)
{
    for (int i = 0; i < 3; ++i) {
        const int a = (4 + 2 * i) * 20 + 10, b = (5 + 2 * i) * 20 + 10;
        for (int o = 0; o < 2; ++o) {
            const int c = 3 * o, d = i * 13 + 4 * o;
            B(d + 0) = -A(b + c);
            B(d + 1) = A(a + c) - A(b + c + 1);
            B(d + 2) = A(a + c + 1) - A(b + c + 2);
            B(d + 3) = A(a + c + 2);
        }
        const int d = i * 13 + 8;
        B(d + 0) = -A(b + 6);
        B(d + 1) = A(a + 6) - A(b + 7);
        B(d + 2) = A(a + 7) - A(b + 8);
        B(d + 3) = A(a + 8) - A(b + 9);
        B(d + 4) = A(a + 9);
    }
}

// det B(z) as a polynomial of degree <= 10, coefficients lowest power first in c[0..10].
template <class Mem, class MemC>
GMS_HD void determinant_poly(const Mem& B, MemC& c)
{
    // entry (i, j) has degree deg_j = {3, 3, 4}; p(i, j, k) = coefficient of z^k
    auto p = [&](int i, int j, int k) -> double {
        const int deg = j == 2 ? 4 : 3;
        return (k < 0 || k > deg) ? 0.0 : B(i * 13 + 4 * j + (deg - k));
    };
    for (int k = 0; k <= 10; ++k) c(k) = 0.0;
    const int perm[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};  // column of rows 0, 1, 2
    for (int t = 0; t < 6; ++t) {
        const double sg = t < 3 ? 1.0 : -1.0;
        const int j0 = perm[t][0], j1 = perm[t][1], j2 = perm[t][2];
        const int d0 = j0 == 2 ? 4 : 3, d1 = j1 == 2 ? 4 : 3, d2 = j2 == 2 ? 4 : 3;
        for (int a = 0; a <= d0; ++a) {
            const double pa = sg * p(0, j0, a);
            if (pa == 0.0) continue;
            for (int b = 0; b <= d1; ++b) {
                const double pab = pa * p(1, j1, b);
                for (int e = 0; e <= d2; ++e) c(a + b + e) += pab * p(2, j2, e);
            }
        }
    }
}

// All roots of a real polynomial of degree n <= 10 (c lowest power first, c(n) != 0) by Aberth's simultaneous iteration in complex
// fp64; re / im receive n roots. Real roots are polished by Newton afterwards (caller). The arrays are indexed at run time: they
// live behind accessors (LDS on the device: a lane's private arrays would sit in scratch memory, a round trip per access).
template <class MemC, class MemR>
GMS_HD int aberth_roots(const MemC& c, int n, MemR& re, MemR& im)
{
    const double an = c(n);
    // Fujiwara's bound on the root moduli
    double radius = 0.0;
    for (int k = 1; k <= n; ++k) {
        const double q = fabs(c(n - k) / an) * (k == n ? 0.5 : 1.0);
        if (q > 0.0) radius = fmax(radius, pow(q, 1.0 / k));
    }
    radius = 2.0 * radius;
    if (!(radius > 0.0)) radius = 1.0;
    const double centre = -c(n - 1) / (an * n);
    for (int k = 0; k < n; ++k) {
        const double ang = 6.283185307179586 * k / n + 0.4;
        re(k) = centre + 0.5 * radius * cos(ang);
        im(k) = 0.5 * radius * sin(ang);
    }
    int it = 0;
    for (; it < 48; ++it) {  // (cubic convergence: a dozen iterations as a rule; the roots that matter are polished afterwards)
        double worst = 0.0;
        for (int i = 0; i < n; ++i) {
            const double zr = re(i), zi = im(i);
            // p(z), p'(z) by Horner
            double pr = an, pi = 0.0, dr = 0.0, di = 0.0;
            for (int k = n - 1; k >= 0; --k) {
                const double ndr = dr * zr - di * zi + pr, ndi = dr * zi + di * zr + pi;
                dr = ndr;
                di = ndi;
                const double npr = pr * zr - pi * zi + c(k), npi = pr * zi + pi * zr;
                pr = npr;
                pi = npi;
            }
            const double dd = dr * dr + di * di;
            if (dd == 0.0) {  // on a critical point: nudge
                re(i) = zr + 1e-8 * (1.0 + fabs(zr));
                worst = 1.0;
                continue;
            }
            // w = p / p'
            const double wr = (pr * dr + pi * di) / dd, wi = (pi * dr - pr * di) / dd;
            // s = sum over j != i of 1 / (z_i - z_j)
            double sr = 0.0, si = 0.0;
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                const double er = zr - re(j), ei = zi - im(j);
                const double ee = er * er + ei * ei;
                if (ee == 0.0) continue;
                sr += er / ee;
                si -= ei / ee;
            }
            // step = w / (1 - w s)
            const double qr = 1.0 - (wr * sr - wi * si), qi = -(wr * si + wi * sr);
            const double qq = qr * qr + qi * qi;
            double stepr = wr, stepi = wi;
            if (qq > 0.0) {
                stepr = (wr * qr + wi * qi) / qq;
                stepi = (wi * qr - wr * qi) / qq;
            }
            re(i) = zr - stepr;
            im(i) = zi - stepi;
            const double rel = sqrt(stepr * stepr + stepi * stepi) / fmax(1.0, sqrt(zr * zr + zi * zi));
            worst = fmax(worst, rel);
        }
        if (worst < 1e-12) break;
    }
    return it;
}

// The same iteration for the degree that always occurs in practice (10: the determinant's leading coefficient vanishes only on
// degenerate samples), on plain arrays with every index a compile-time constant: on the device the coefficients and the roots then
// live in registers instead of LDS (the generic form pays an LDS round trip per operand, and sixteen samples in lockstep run as
// many sweeps as the slowest of them -- 48 more often than not). Statement for statement the generic loop: same operations in the
// same order, bit-identical results.
GMS_HD int aberth_roots_10(const double (&c)[11], double (&re)[10], double (&im)[10])
{
    constexpr int n = 10;
    const double an = c[n];
    double radius = 0.0;
GMS_UNROLL
    for (int k = 1; k <= n; ++k) {
        const double q = fabs(c[n - k] / an) * (k == n ? 0.5 : 1.0);
        if (q > 0.0) radius = fmax(radius, pow(q, 1.0 / k));
    }
    radius = 2.0 * radius;
    if (!(radius > 0.0)) radius = 1.0;
    const double centre = -c[n - 1] / (an * n);
GMS_UNROLL
    for (int k = 0; k < n; ++k) {
        const double ang = 6.283185307179586 * k / n + 0.4;
        re[k] = centre + 0.5 * radius * cos(ang);
        im[k] = 0.5 * radius * sin(ang);
    }
    int it = 0;
    for (; it < 48; ++it) {
        double worst = 0.0;
GMS_UNROLL
        for (int i = 0; i < n; ++i) {
            const double zr = re[i], zi = im[i];
            double pr = an, pi = 0.0, dr = 0.0, di = 0.0;
GMS_UNROLL
            for (int k = n - 1; k >= 0; --k) {
                const double ndr = dr * zr - di * zi + pr, ndi = dr * zi + di * zr + pi;
                dr = ndr;
                di = ndi;
                const double npr = pr * zr - pi * zi + c[k], npi = pr * zi + pi * zr;
                pr = npr;
                pi = npi;
            }
            const double dd = dr * dr + di * di;
            if (dd == 0.0) {  // on a critical point: nudge
                re[i] = zr + 1e-8 * (1.0 + fabs(zr));
                worst = 1.0;
                continue;
            }
            const double wr = (pr * dr + pi * di) / dd, wi = (pi * dr - pr * di) / dd;
            double sr = 0.0, si = 0.0;
GMS_UNROLL
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                const double er = zr - re[j], ei = zi - im[j];
                const double ee = er * er + ei * ei;
                if (ee == 0.0) continue;
                sr += er / ee;
                si -= ei / ee;
            }
            const double qr = 1.0 - (wr * sr - wi * si), qi = -(wr * si + wi * sr);
            const double qq = qr * qr + qi * qi;
            double stepr = wr, stepi = wi;
            if (qq > 0.0) {
                stepr = (wr * qr + wi * qi) / qq;
                stepi = (wi * qr - wr * qi) / qq;
            }
            re[i] = zr - stepr;
            im[i] = zi - stepi;
            const double rel = sqrt(stepr * stepr + stepi * stepi) / fmax(1.0, sqrt(zr * zr + zi * zi));
            worst = fmax(worst, rel);
        }
        if (worst < 1e-12) break;
    }
    return it;
}

// E with its largest-magnitude entry positive (first such entry on ties)
GMS_HD void canonical_sign(double* E)
{
    int k = 0;
    for (int i = 1; i < 9; ++i)
        if (fabs(E[i]) > fabs(E[k])) k = i;
    if (E[k] < 0.0)
        for (int i = 0; i < 9; ++i) E[i] = -E[i];
}

// Polish one solution (x, y, z) of the ten constraints by Gauss-Newton on the constraints themselves, evaluated from E = x X + y Y +
// z Z + W directly (the eliminated system and the degree-10 root carry the conditioning of the elimination; the constraints do not).
// r = [det E, vec(2 E E^T E - tr(E E^T) E)], dr[D] = [tr(adj(E) D), vec(2 (D E^T E + E D^T E + E E^T D) - 2 tr(E D^T) E - tr(E E^T) D)].
// A step (halved up to seven times if need be) is taken only while it lowers |r|; at most ten steps.
template <class Mem>
GMS_HD void polish_solution(const Mem& basis, double& x, double& y, double& z)
{
    auto eval = [&](double px, double py, double pz, double E[9], double r[10]) {
        for (int k = 0; k < 9; ++k) E[k] = px * basis(k) + py * basis(9 + k) + pz * basis(18 + k) + basis(27 + k);
        double G[9];  // E E^T
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) G[3 * a + b] = E[3 * a] * E[3 * b] + E[3 * a + 1] * E[3 * b + 1] + E[3 * a + 2] * E[3 * b + 2];
        const double tr = G[0] + G[4] + G[8];
        r[0] = E[0] * (E[4] * E[8] - E[5] * E[7]) - E[1] * (E[3] * E[8] - E[5] * E[6]) + E[2] * (E[3] * E[7] - E[4] * E[6]);
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                r[1 + 3 * a + b] = 2.0 * (G[3 * a] * E[b] + G[3 * a + 1] * E[3 + b] + G[3 * a + 2] * E[6 + b]) - tr * E[3 * a + b];
    };
    double E[9], r[10];
#if defined(TV_DIAG) && TV_DIAG == 1   // (timing-only diagnostic builds, tools/fivepoint_bench.py: of 9 ms per 16 384 samples the root finder is 3.4, the polish 0.7)
    return;
#endif
    eval(x, y, z, E, r);
    double r2 = 0.0;
    for (int k = 0; k < 10; ++k) r2 += r[k] * r[k];
    for (int it = 0; it < 10 && r2 > 0.0; ++it) {
        // Jacobian columns: D = X, Y, Z
        double J[3][10];
        double G[9], EtE[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                G[3 * a + b] = E[3 * a] * E[3 * b] + E[3 * a + 1] * E[3 * b + 1] + E[3 * a + 2] * E[3 * b + 2];      // E E^T
                EtE[3 * a + b] = E[a] * E[b] + E[3 + a] * E[3 + b] + E[6 + a] * E[6 + b];                            // E^T E
            }
        const double tr = G[0] + G[4] + G[8];
        const double adj[9] = {E[4] * E[8] - E[5] * E[7], E[2] * E[7] - E[1] * E[8], E[1] * E[5] - E[2] * E[4],
                               E[5] * E[6] - E[3] * E[8], E[0] * E[8] - E[2] * E[6], E[2] * E[3] - E[0] * E[5],
                               E[3] * E[7] - E[4] * E[6], E[1] * E[6] - E[0] * E[7], E[0] * E[4] - E[1] * E[3]};
        for (int v = 0; v < 3; ++v) {
            double D[9];
            for (int k = 0; k < 9; ++k) D[k] = basis(9 * v + k);
            // tr(adj(E) D)
            double d0 = 0.0;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) d0 += adj[3 * a + b] * D[3 * b + a];
            J[v][0] = d0;
            double trED = 0.0;  // tr(E D^T)
            for (int k = 0; k < 9; ++k) trED += E[k] * D[k];
            double DtE[9];  // D^T E
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) DtE[3 * a + b] = D[a] * E[b] + D[3 + a] * E[3 + b] + D[6 + a] * E[6 + b];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double t = 0.0;
                    for (int k = 0; k < 3; ++k) t += D[3 * a + k] * EtE[3 * k + b] + E[3 * a + k] * DtE[3 * k + b] + G[3 * a + k] * D[3 * k + b];
                    J[v][1 + 3 * a + b] = 2.0 * t - 2.0 * trED * E[3 * a + b] - tr * D[3 * a + b];
                }
        }
        // normal equations (3 x 3), Cramer
        double N[3][3], g[3];
        for (int a = 0; a < 3; ++a) {
            g[a] = 0.0;
            for (int k = 0; k < 10; ++k) g[a] += J[a][k] * r[k];
            for (int b = 0; b < 3; ++b) {
                N[a][b] = 0.0;
                for (int k = 0; k < 10; ++k) N[a][b] += J[a][k] * J[b][k];
            }
        }
        const double c00 = N[1][1] * N[2][2] - N[1][2] * N[2][1], c01 = N[1][2] * N[2][0] - N[1][0] * N[2][2], c02 = N[1][0] * N[2][1] - N[1][1] * N[2][0];
        const double detN = N[0][0] * c00 + N[0][1] * c01 + N[0][2] * c02;
        if (!(fabs(detN) > 0.0) || !isfinite(detN)) break;
        const double dx = -(g[0] * c00 + g[1] * (N[0][2] * N[2][1] - N[0][1] * N[2][2]) + g[2] * (N[0][1] * N[1][2] - N[0][2] * N[1][1])) / detN;
        const double dy = -(g[0] * c01 + g[1] * (N[0][0] * N[2][2] - N[0][2] * N[2][0]) + g[2] * (N[0][2] * N[1][0] - N[0][0] * N[1][2])) / detN;
        const double dz = -(g[0] * c02 + g[1] * (N[0][1] * N[2][0] - N[0][0] * N[2][1]) + g[2] * (N[0][0] * N[1][1] - N[0][1] * N[1][0])) / detN;
        if (!isfinite(dx) || !isfinite(dy) || !isfinite(dz)) break;
        double En[9], rn[10], rn2 = 0.0, f = 1.0;
        bool better = false;
        for (int half = 0; half < 8 && !better; ++half, f *= 0.5) {  // the full step, else halved until the residual drops
            eval(x + f * dx, y + f * dy, z + f * dz, En, rn);
            rn2 = 0.0;
            for (int k = 0; k < 10; ++k) rn2 += rn[k] * rn[k];
            better = rn2 < r2;
            if (better) {
                x += f * dx;
                y += f * dy;
                z += f * dz;
            }
        }
        if (!better) break;
        r2 = rn2;
        for (int k = 0; k < 9; ++k) E[k] = En[k];
        for (int k = 0; k < 10; ++k) r[k] = rn[k];
    }
}

// Everything of one sample: models[9 * k] receives the k-th essential matrix (row-major, unit Frobenius norm, largest entry
// positive, x2^T E x1 = 0); returns their number (0..10). The order of a sample's models decides RANSAC ties and must not depend on
// the null-space basis (the roots z do): ascending in E[0], then E[1], ...
template <class Mem, class MemOut>
GMS_HD int five_point(const double x1[5], const double y1[5], const double x2[5], const double y2[5], FivePointMem<Mem>& m, MemOut& models)
{
    if (!null_space_5x9(x1, y1, x2, y2, m.basis, m.work)) return 0;
#if defined(TV_DIAG) && TV_DIAG == 4
    return 0;
#endif
    constraint_matrix(m.basis, m.A, m.work);
#if defined(TV_DIAG) && TV_DIAG == 5
    return 0;
#endif
    if (!eliminate_10x20(m.A)) return 0;
#if defined(TV_DIAG) && TV_DIAG == 6
    return 0;
#endif
    reduced_system(m.A, m.work);  // B = work[0..38]
    // the constraint matrix has done its work: its storage now holds the run-time indexed arrays of the root finder
    struct Sub {
        Mem* base;
        int off;
        GMS_HD double& operator()(int i) const { return (*base)(off + i); }
    };
    Sub c{&m.A, 0}, re{&m.A, 16}, im{&m.A, 32}, zs{&m.A, 48};
    determinant_poly(m.work, c);
    double cmax = 0.0;
    bool finite = true;
    for (int k = 0; k <= 10; ++k) {
        finite = finite && isfinite(c(k));
        cmax = fmax(cmax, fabs(c(k)));
    }
    if (!finite || cmax == 0.0) return 0;
    int n = 10;
    while (n > 0 && c(n) == 0.0) --n;
    if (n == 0) return 0;
#if defined(TV_DIAG) && TV_DIAG == 2
    return 0;
#endif
    if (n == 10) {
        double cl[11], rl[10], il[10];
GMS_UNROLL
        for (int k = 0; k <= 10; ++k) cl[k] = c(k);
        aberth_roots_10(cl, rl, il);
GMS_UNROLL
        for (int k = 0; k < 10; ++k) {
            re(k) = rl[k];
            im(k) = il[k];
        }
    } else {
        aberth_roots(c, n, re, im);
    }
#if defined(TV_DIAG) && TV_DIAG == 3
    return 0;
#endif
    // the real roots (|imag| < 1e-10), polished by two Newton steps on the real polynomial
    int nz = 0;
    for (int i = 0; i < n; ++i) {
        if (!(fabs(im(i)) < 1e-10) || !isfinite(re(i))) continue;
        double z = re(i);
        for (int s = 0; s < 2; ++s) {
            double pv = c(n), dv = 0.0;
            for (int k = n - 1; k >= 0; --k) {
                dv = dv * z + pv;
                pv = pv * z + c(k);
            }
            if (dv != 0.0 && isfinite(pv / dv)) z -= pv / dv;
        }
        int pos = nz;
        while (pos > 0 && zs(pos - 1) > z) {
            zs(pos) = zs(pos - 1);
            --pos;
        }
        zs(pos) = z;
        ++nz;
    }
    int count = 0;
    for (int i = 0; i < nz; ++i) {
        const double z = zs(i);
        // B(z): 3 x 3; its null vector (x, y, 1) up to scale = the largest of the three row cross products
        double bz[3][3];
        for (int r = 0; r < 3; ++r) {
            const int o = r * 13;
            bz[r][0] = ((m.work(o + 0) * z + m.work(o + 1)) * z + m.work(o + 2)) * z + m.work(o + 3);
            bz[r][1] = ((m.work(o + 4) * z + m.work(o + 5)) * z + m.work(o + 6)) * z + m.work(o + 7);
            bz[r][2] = (((m.work(o + 8) * z + m.work(o + 9)) * z + m.work(o + 10)) * z + m.work(o + 11)) * z + m.work(o + 12);
        }
        double v[3] = {0.0, 0.0, 0.0}, best = -1.0;
        for (int t = 0; t < 3; ++t) {
            const int a = t == 2 ? 0 : t, b = t == 0 ? 1 : 2;  // (0,1), (1,2), (0,2)
            const double cx = bz[a][1] * bz[b][2] - bz[a][2] * bz[b][1];
            const double cy = bz[a][2] * bz[b][0] - bz[a][0] * bz[b][2];
            const double cz = bz[a][0] * bz[b][1] - bz[a][1] * bz[b][0];
            const double n2 = cx * cx + cy * cy + cz * cz;
            if (n2 > best) {
                best = n2;
                v[0] = cx;
                v[1] = cy;
                v[2] = cz;
            }
        }
        if (!(best > 0.0)) continue;
        const double vn = sqrt(best);
        if (fabs(v[2]) < 1e-10 * vn) continue;  // (SVD::solveZ returns a unit vector: |v_2| < 1e-10 there)
        double x = v[0] / v[2], y = v[1] / v[2], zp = z;
        polish_solution(m.basis, x, y, zp);
        double e[9], n2 = 0.0;
        for (int k = 0; k < 9; ++k) {
            e[k] = x * m.basis(k) + y * m.basis(9 + k) + zp * m.basis(18 + k) + m.basis(27 + k);
            n2 += e[k] * e[k];
        }
        if (!(n2 > 0.0) || !isfinite(n2)) continue;
        const double inv = 1.0 / sqrt(n2);
        for (int k = 0; k < 9; ++k) e[k] *= inv;
        canonical_sign(e);
        // insertion into the ordered list
        int pos = count;
        while (pos > 0) {
            bool less = false;  // e < models[pos - 1] lexicographically
            for (int k = 0; k < 9; ++k) {
                const double prev = models(9 * (pos - 1) + k);
                if (e[k] != prev) {
                    less = e[k] < prev;
                    break;
                }
            }
            if (!less) break;
            for (int k = 0; k < 9; ++k) models(9 * pos + k) = models(9 * (pos - 1) + k);
            --pos;
        }
        for (int k = 0; k < 9; ++k) models(9 * pos + k) = e[k];
        ++count;
    }
    return count;
}

// ---- cv::decomposeEssentialMat: E = U diag(s, s, 0) V^T with det U = det V = +1, R1 = U W V^T, R2 = U W^T V^T, t = U's last column.
// V and the singular values from the eigen-decomposition of E^T E (cyclic Jacobi), U's first two columns from E v / |E v|
// (re-orthogonalised), its third as their cross product (det U = +1 by construction). false: rank below two.
GMS_HD bool decompose_essential(const double E[9], double R1[9], double R2[9], double t[3])
{
    double S[3][3], V[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            S[a][b] = E[a] * E[b] + E[3 + a] * E[3 + b] + E[6 + a] * E[6 + b];
            V[a][b] = a == b ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (S[p][q] == 0.0) continue;
                const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int k = 0; k < 3; ++k) {
                    const double a = S[k][p], b = S[k][q];
                    S[k][p] = cs * a - sn * b;
                    S[k][q] = sn * a + cs * b;
                }
                for (int k = 0; k < 3; ++k) {
                    const double a = S[p][k], b = S[q][k];
                    S[p][k] = cs * a - sn * b;
                    S[q][k] = sn * a + cs * b;
                }
                for (int k = 0; k < 3; ++k) {
                    const double a = V[k][p], b = V[k][q];
                    V[k][p] = cs * a - sn * b;
                    V[k][q] = sn * a + cs * b;
                }
            }
    }
    int ord[3] = {0, 1, 2};  // eigenvalues descending
    for (int a = 0; a < 3; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (S[ord[b]][ord[b]] > S[ord[a]][ord[a]]) {
                const int tmp = ord[a];
                ord[a] = ord[b];
                ord[b] = tmp;
            }
    double v[3][3];  // v[j] = j-th right singular vector
    for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) v[j][k] = V[k][ord[j]];
    const double det = v[0][0] * (v[1][1] * v[2][2] - v[1][2] * v[2][1]) - v[0][1] * (v[1][0] * v[2][2] - v[1][2] * v[2][0]) +
                       v[0][2] * (v[1][0] * v[2][1] - v[1][1] * v[2][0]);
    if (det < 0.0)
        for (int k = 0; k < 3; ++k) v[2][k] = -v[2][k];  // (the null direction: E v2 = 0 either way)
    double u[3][3];
    for (int j = 0; j < 2; ++j) {
        double n2 = 0.0;
        for (int r = 0; r < 3; ++r) {
            u[j][r] = E[3 * r] * v[j][0] + E[3 * r + 1] * v[j][1] + E[3 * r + 2] * v[j][2];
            n2 += u[j][r] * u[j][r];
        }
        if (!(n2 > 0.0)) return false;
        const double inv = 1.0 / sqrt(n2);
        for (int r = 0; r < 3; ++r) u[j][r] *= inv;
    }
    {
        const double d = u[0][0] * u[1][0] + u[0][1] * u[1][1] + u[0][2] * u[1][2];
        double n2 = 0.0;
        for (int r = 0; r < 3; ++r) {
            u[1][r] -= d * u[0][r];
            n2 += u[1][r] * u[1][r];
        }
        if (!(n2 > 0.0)) return false;
        const double inv = 1.0 / sqrt(n2);
        for (int r = 0; r < 3; ++r) u[1][r] *= inv;
    }
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    // W = [0 1 0; -1 0 0; 0 0 1]: U W = [-u1, u0, u2], U W^T = [u1, -u0, u2] (columns)
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            R1[3 * r + c] = -u[1][r] * v[0][c] + u[0][r] * v[1][c] + u[2][r] * v[2][c];
            R2[3 * r + c] = u[1][r] * v[0][c] - u[0][r] * v[1][c] + u[2][r] * v[2][c];
        }
    for (int r = 0; r < 3; ++r) t[r] = u[2][r];
    return true;
}

// ---- cv::undistortPoints (five fixed-point iterations of the distortion model) and cv::triangulatePoints' DLT ---------------------------
struct Camera {
    double fx, fy, cx, cy, k1, k2, p1, p2, k3;
};

GMS_HD void undistort_point(const Camera& c, double u, double v, double& xo, double& yo)
{
    const double x0 = (u - c.cx) / c.fx, y0 = (v - c.cy) / c.fy;
    double x = x0, y = y0;
    if (c.k1 != 0.0 || c.k2 != 0.0 || c.p1 != 0.0 || c.p2 != 0.0 || c.k3 != 0.0) {
        for (int it = 0; it < 5; ++it) {
            const double r2 = x * x + y * y;
            const double icdist = 1.0 / (1.0 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
            const double dx = 2.0 * c.p1 * x * y + c.p2 * (r2 + 2.0 * x * x);
            const double dy = c.p1 * (r2 + 2.0 * y * y) + 2.0 * c.p2 * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
    }
    xo = x;
    yo = y;
}

// The homogeneous X with x1 ~ Pa X, x2 ~ Pb X: the right singular vector of the smallest singular value of the 4 x 4 DLT matrix,
// as the eigenvector of the smallest eigenvalue of A^T A (cyclic Jacobi, fp64).
GMS_HD void dlt_point(const double* Pa, const double* Pb, double x1, double y1, double x2, double y2, double X[4])
{
    double A[4][4];
GMS_UNROLL
    for (int k = 0; k < 4; ++k) {
        A[0][k] = x1 * Pa[8 + k] - Pa[k];
        A[1][k] = y1 * Pa[8 + k] - Pa[4 + k];
        A[2][k] = x2 * Pb[8 + k] - Pb[k];
        A[3][k] = y2 * Pb[8 + k] - Pb[4 + k];
    }
    double S[4][4], V[4][4];   // (every index below is a compile-time constant once the loops are unrolled: registers on the device)
GMS_UNROLL
    for (int a = 0; a < 4; ++a)
GMS_UNROLL
        for (int b = 0; b < 4; ++b) {
            S[a][b] = A[0][a] * A[0][b] + A[1][a] * A[1][b] + A[2][a] * A[2][b] + A[3][a] * A[3][b];
            V[a][b] = a == b ? 1.0 : 0.0;
        }
    // sweeps until the off-diagonal mass is below rounding relative to the diagonal (it falls quadratically: five or six sweeps)
    const double tr0 = S[0][0] + S[1][1] + S[2][2] + S[3][3];
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0;
GMS_UNROLL
        for (int a = 0; a < 4; ++a)
GMS_UNROLL
            for (int b = a + 1; b < 4; ++b) off += S[a][b] * S[a][b];
        if (off <= 1e-36 * tr0 * tr0) break;
GMS_UNROLL
        for (int p = 0; p < 3; ++p)
GMS_UNROLL
            for (int q = p + 1; q < 4; ++q) {
                if (S[p][q] == 0.0) continue;
                const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
GMS_UNROLL
                for (int k = 0; k < 4; ++k) {
                    const double skp = S[k][p], skq = S[k][q];
                    S[k][p] = cs * skp - sn * skq;
                    S[k][q] = sn * skp + cs * skq;
                }
GMS_UNROLL
                for (int k = 0; k < 4; ++k) {
                    const double spk = S[p][k], sqk = S[q][k];
                    S[p][k] = cs * spk - sn * sqk;
                    S[q][k] = sn * spk + cs * sqk;
                }
GMS_UNROLL
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = cs * vkp - sn * vkq;
                    V[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    int best = 0;
    double dmin = S[0][0];   // (the smallest diagonal entry, the first one on ties; no run-time index into S)
GMS_UNROLL
    for (int k = 1; k < 4; ++k)
        if (S[k][k] < dmin) {
            dmin = S[k][k];
            best = k;
        }
GMS_UNROLL
    for (int k = 0; k < 4; ++k) X[k] = best == 0 ? V[k][0] : best == 1 ? V[k][1] : best == 2 ? V[k][2] : V[k][3];
}

// cv::recoverPose's cheirality test of one correspondence under the four candidate poses (R1, t), (R2, t), (R1, -t), (R2, -t): bit h set
// = positive depth below the threshold in both cameras. P[h] = [R | t] row-major 3 x 4, P[h + 2] = [R | -t]. The DLT matrix of (R, -t)
// is that of (R, t) with its last column negated, so its null vector is (X, -w): two decompositions serve four hypotheses.
GMS_HD unsigned pose_votes(const double P[4][12], double dist_thresh, double x1, double y1, double x2, double y2)
{
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    unsigned bits = 0;
    for (int h = 0; h < 2; ++h) {
        double Q[4];
        dlt_point(P0, P[h], x1, y1, x2, y2, Q);
        const double qx = Q[0] / Q[3], qy = Q[1] / Q[3], qz = Q[2] / Q[3];
        const double z2 = P[h][8] * qx + P[h][9] * qy + P[h][10] * qz + P[h][11];
        // (R, t): Q as it stands
        if (Q[2] * Q[3] > 0.0 && qz < dist_thresh && z2 > 0.0 && z2 < dist_thresh) bits |= 1u << h;
        // (R, -t): (X, -w) -- the point mirrored through the first camera, the second depth negated
        if (-(Q[2] * Q[3]) > 0.0 && -qz < dist_thresh && -z2 > 0.0 && -z2 < dist_thresh) bits |= 1u << (h + 2);
    }
    return bits;
}

// the same test for ONE hypothesis (the winner, when the mask is written)
GMS_HD bool pose_vote_one(const double P[4][12], int h, double dist_thresh, double x1, double y1, double x2, double y2)
{
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double Q[4];
    dlt_point(P0, P[h & 1], x1, y1, x2, y2, Q);
    const double sg = h < 2 ? 1.0 : -1.0;
    const double qz = sg * (Q[2] / Q[3]);
    const double z2 = sg * (P[h & 1][8] * (Q[0] / Q[3]) + P[h & 1][9] * (Q[1] / Q[3]) + P[h & 1][10] * (Q[2] / Q[3]) + P[h & 1][11]);
    return sg * (Q[2] * Q[3]) > 0.0 && qz < dist_thresh && z2 > 0.0 && z2 < dist_thresh;
}

}  // namespace tv
}  // namespace gms
