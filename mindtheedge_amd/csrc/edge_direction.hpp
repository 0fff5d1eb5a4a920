// Direction bin of an edge-normal angle -- GradLayer's masks, packnet_sfm/losses/grad_loss.py:80-93: thresholds float32(k*pi/8), half-open
// bins [lo, hi), later assignments win; 0: h, 1: v, 2: rl, 3: lr.  Plain C++ so that the CPU suite can check the fast form against the
// literal one over ALL 2^32 float patterns (tests/test_edge_direction_code.py compiles this header with g++).
#pragma once
#include <cstring>
#if defined(__HIPCC__)
#define MTE_EDGE_FN __device__ __forceinline__
#else
#define MTE_EDGE_FN static inline
#endif

// the reference's masks, literally
MTE_EDGE_FN int direction_code_literal(float n) {
    const float P1 = (float)(1 * 3.14159265358979323846 / 8), P3 = (float)(3 * 3.14159265358979323846 / 8),
                P5 = (float)(5 * 3.14159265358979323846 / 8), P7 = (float)(7 * 3.14159265358979323846 / 8);
    int code = 0;                                                   // 0: h
    if ((n >= -P5 && n < -P3) || (n >= P3 && n < P5)) code = 1;     // v
    if ((n >= -P7 && n < -P5) || (n >= P1 && n < P3)) code = 2;     // rl
    if ((n >= -P3 && n < -P1) || (n >= P5 && n < P7)) code = 3;     // lr
    return code;
}

// The same function in ~14 instructions instead of ~37 (eight float compares + mask logic): non-negative floats order like their bit
// patterns, so k = how many of the four thresholds |n| has reached is four integer compares; on the negative side the bins are closed at
// the OTHER end ([-P3, -P1) = |n| in (P1, P3]), i.e. "|n| > P" = "bits(|n|) - 1 >= bits(P)"; the code is a 2-bit field of a per-sign table.
// -0.0, NaN and +-inf fall out right (k = 0 or 4 -> code 0, as no literal bin holds them).
MTE_EDGE_FN int direction_code(float n) {
    const int p1 = 0x3EC90FDB, p3 = 0x3F96CBE4, p5 = 0x3FFB53D1, p7 = 0x402FEDDF;     // float32(pi/8), (3pi/8), (5pi/8), (7pi/8)
    unsigned u;
#if defined(__HIPCC__)
    u = __builtin_bit_cast(unsigned, n);
#else
    std::memcpy(&u, &n, 4);
#endif
    const int neg = (int)(u >> 31);
    const int mi = (int)(u & 0x7fffffffu) - neg;
    const int k = (mi >= p1) + (mi >= p3) + (mi >= p5) + (mi >= p7);
    const unsigned tbl = neg ? 0x9Cu : 0xD8u;                       // k = 0..4 -> {h, lr, v, rl, h} below zero, {h, rl, v, lr, h} above
    return (int)((tbl >> (2 * k)) & 3u);
}

// The same bins as three range tests + the sign (what the fast loss kernels consume: they select among the four Sobel responses and build the
// backward planes straight from these masks, never forming the code): k1 = |n| in the first bin next to zero, v = the middle bin, k3 = the
// bin next to +-pi; on the positive side k1 is rl and k3 is lr, below zero the other way round.
struct DirMasks { bool v, k1, k3, neg; };
MTE_EDGE_FN DirMasks direction_masks(float n) {
    const int p1 = 0x3EC90FDB, p3 = 0x3F96CBE4, p5 = 0x3FFB53D1, p7 = 0x402FEDDF;
    unsigned u;
#if defined(__HIPCC__)
    u = __builtin_bit_cast(unsigned, n);
#else
    std::memcpy(&u, &n, 4);
#endif
    const int mi = (int)(u & 0x7fffffffu) - (int)(u >> 31);
    DirMasks m;
    m.neg = (int)u < 0;
    m.k1 = (unsigned)(mi - p1) < (unsigned)(p3 - p1);
    m.v = (unsigned)(mi - p3) < (unsigned)(p5 - p3);
    m.k3 = (unsigned)(mi - p5) < (unsigned)(p7 - p5);
    return m;
}
MTE_EDGE_FN int code_from_masks(DirMasks m) { return m.v ? 1 : (m.k1 ? (m.neg ? 3 : 2) : (m.k3 ? (m.neg ? 2 : 3) : 0)); }
