// Deterministic float32 math shared by the HIP kernels and the C oracle build.
//
// Every operation here is a single IEEE-754 correctly-rounded float32 (or float64)
// operation in a fixed order: no FMA contraction, no library calls whose results
// differ between hosts.  Compiled for the device by hipcc (explicit __f*_rn
// intrinsics) and for the host by gcc/clang with -ffp-contract=off, the functions
// return bit-identical results -- which is what makes "HIP dibits bit-exact vs the
// CPU restatement" (SURVEY.md a') attainable through the C4FM feedback loop.
#pragma once

#if defined(__HIP_DEVICE_COMPILE__)
#define WHM_FN __device__ __forceinline__
#define WHM_MUL(a, b) __fmul_rn((a), (b))
#define WHM_ADD(a, b) __fadd_rn((a), (b))
#define WHM_SUB(a, b) __fsub_rn((a), (b))
#define WHM_DIV(a, b) __fdiv_rn((a), (b))
#define WHM_DMUL(a, b) __dmul_rn((a), (b))
#define WHM_DADD(a, b) __dadd_rn((a), (b))
#define WHM_DSUB(a, b) __dsub_rn((a), (b))
#elif defined(__HIPCC__)
#define WHM_FN __host__ __device__ inline
#define WHM_MUL(a, b) ((a) * (b))
#define WHM_ADD(a, b) ((a) + (b))
#define WHM_SUB(a, b) ((a) - (b))
#define WHM_DIV(a, b) ((a) / (b))
#define WHM_DMUL(a, b) ((a) * (b))
#define WHM_DADD(a, b) ((a) + (b))
#define WHM_DSUB(a, b) ((a) - (b))
#else
#define WHM_FN static inline
#define WHM_MUL(a, b) ((a) * (b))
#define WHM_ADD(a, b) ((a) + (b))
#define WHM_SUB(a, b) ((a) - (b))
#define WHM_DIV(a, b) ((a) / (b))
#define WHM_DMUL(a, b) ((a) * (b))
#define WHM_DADD(a, b) ((a) + (b))
#define WHM_DSUB(a, b) ((a) - (b))
#endif

// atan2f, max error ~2 ulp.  Octant reduction + cephes atanf minimax polynomial.
// Signed zeros follow C / IEEE atan2f (and numpy's arctan2): the quadrant is taken from the SIGN BITS, so
// atan2f(+-0, -0) = +-pi, atan2f(+-0, +0) = +-0, atan2f(-0, x < 0) = -pi -- a muted input (exact zeros through the filters)
// reaches the C4FM discriminator as products of signed zeros, and the reference's phase there is 0 or pi by their signs.
WHM_FN float whm_atan2f(float y, float x) {
    const float PI_F = 3.14159265358979323846f;
    const float PIO2_F = 1.57079632679489661923f;
    const float PIO4_F = 0.78539816339744830962f;
    float ax = x < 0.0f ? -x : x;
    float ay = y < 0.0f ? -y : y;
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    float r;
    if (mx == 0.0f) {
        r = 0.0f;
    } else {
        float t = WHM_DIV(mn, mx);  // [0, 1]
        float base = 0.0f;
        if (t > 0.4142135623730950f) {  // tan(pi/8)
            t = WHM_DIV(WHM_SUB(t, 1.0f), WHM_ADD(t, 1.0f));
            base = PIO4_F;
        }
        float z = WHM_MUL(t, t);
        float p = 8.05374449538e-2f;
        p = WHM_SUB(WHM_MUL(p, z), 1.38776856032e-1f);
        p = WHM_ADD(WHM_MUL(p, z), 1.99777106478e-1f);
        p = WHM_SUB(WHM_MUL(p, z), 3.33329491539e-1f);
        p = WHM_ADD(WHM_MUL(WHM_MUL(p, z), t), t);
        r = WHM_ADD(base, p);
        if (ay > ax) r = WHM_SUB(PIO2_F, r);
    }
    if (__builtin_signbit(x)) r = WHM_SUB(PI_F, r);
    return __builtin_signbit(y) ? -r : r;
}

// sin/cos of a float32 phase given in radians, |phase| up to ~1e7: float64 range
// reduction to [-pi/4, pi/4], then cephes sinf/cosf polynomials (~1 ulp).
WHM_FN void whm_sincos_phase(float phase, float *s_out, float *c_out) {
    double t = WHM_DMUL((double)phase, 0.15915494309189533577);  // revolutions
    double r = __builtin_rint(t);
    t = WHM_DSUB(t, r);                                          // [-0.5, 0.5]
    double q = __builtin_rint(WHM_DMUL(t, 4.0));                 // -2..2
    double u = WHM_DSUB(t, WHM_DMUL(q, 0.25));                   // [-1/8, 1/8]
    float th = (float)WHM_DMUL(u, 6.28318530717958647692);
    int qi = ((int)q) & 3;
    float z = WHM_MUL(th, th);
    float sp = -1.9515295891e-4f;
    sp = WHM_ADD(WHM_MUL(sp, z), 8.3321608736e-3f);
    sp = WHM_SUB(WHM_MUL(sp, z), 1.6666654611e-1f);
    sp = WHM_ADD(WHM_MUL(WHM_MUL(sp, z), th), th);
    float cp = 2.443315711809948e-5f;
    cp = WHM_SUB(WHM_MUL(cp, z), 1.388731625493765e-3f);
    cp = WHM_ADD(WHM_MUL(cp, z), 4.166664568298827e-2f);
    cp = WHM_ADD(WHM_SUB(WHM_MUL(WHM_MUL(cp, z), z), WHM_MUL(0.5f, z)), 1.0f);
    float s, c;
    switch (qi) {
        case 0: s = sp; c = cp; break;
        case 1: s = cp; c = -sp; break;
        case 2: s = -sp; c = -cp; break;
        default: s = -cp; c = sp; break;
    }
    *s_out = s;
    *c_out = c;
}

// sin/cos of a float64 angle, |x| < ~1e6 rad: Cody-Waite reduction by pi/2 (33-bit head, so k*head is
// exact for |k| < 2^20), then the cephes double kernels on [-pi/4, pi/4] (error < 1 ulp).  Single rounded
// float64 operations in a fixed order, so host and device agree bit for bit.
WHM_FN void whm_sincos_f64(double x, double *s_out, double *c_out) {
    double k = __builtin_rint(WHM_DMUL(x, 0.63661977236758134308));
    double r = WHM_DSUB(WHM_DSUB(x, WHM_DMUL(k, 1.57079632673412561417e+00)), WHM_DMUL(k, 6.07710050650619224932e-11));
    double z = WHM_DMUL(r, r);
    double sp = 1.58962301576546568060e-10;
    sp = WHM_DSUB(WHM_DMUL(sp, z), 2.50507477628578072866e-8);
    sp = WHM_DADD(WHM_DMUL(sp, z), 2.75573136213857245213e-6);
    sp = WHM_DSUB(WHM_DMUL(sp, z), 1.98412698295895385996e-4);
    sp = WHM_DADD(WHM_DMUL(sp, z), 8.33333333332211858878e-3);
    sp = WHM_DSUB(WHM_DMUL(sp, z), 1.66666666666666307295e-1);
    double s = WHM_DADD(r, WHM_DMUL(WHM_DMUL(r, z), sp));
    double cp = -1.13585365213876817300e-11;
    cp = WHM_DADD(WHM_DMUL(cp, z), 2.08757008419747316778e-9);
    cp = WHM_DSUB(WHM_DMUL(cp, z), 2.75573141792967388112e-7);
    cp = WHM_DADD(WHM_DMUL(cp, z), 2.48015872888517045348e-5);
    cp = WHM_DSUB(WHM_DMUL(cp, z), 1.38888888888730564116e-3);
    cp = WHM_DADD(WHM_DMUL(cp, z), 4.16666666666665929218e-2);
    double c = WHM_DADD(WHM_DSUB(1.0, WHM_DMUL(0.5, z)), WHM_DMUL(WHM_DMUL(z, z), cp));
    long long q = (long long)k & 3;
    double so, co;
    if (q == 0) { so = s; co = c; }
    else if (q == 1) { so = c; co = -s; }
    else if (q == 2) { so = -s; co = -c; }
    else { so = -c; co = s; }
    *s_out = so;
    *c_out = co;
}

// |re + j im| as glibc hypotf computes it: exact float64 squares, one rounded sum, correctly rounded sqrt.
WHM_FN float whm_hypotf(float re, float im) {
    double s = WHM_DADD(WHM_DMUL((double)re, (double)re), WHM_DMUL((double)im, (double)im));
#if defined(__HIP_DEVICE_COMPILE__)
    return (float)__dsqrt_rn(s);
#else
    return (float)__builtin_sqrt(s);
#endif
}
