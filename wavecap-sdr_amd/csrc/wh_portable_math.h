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
// Finite inputs only matter on this path; (0,0) -> 0 like C atan2f(+0,+0).
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
    if (x < 0.0f) r = WHM_SUB(PI_F, r);
    return y < 0.0f ? -r : r;
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
