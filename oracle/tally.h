/* tally.h -- TEST INFRASTRUCTURE (oracle/): the op tally SURVEY.md section 8(d) asks for ("exact count from the oracle's op tally"),
 * without touching a line of the restatement.  oracle/r3o_tally.cpp compiles r3o.c AS C++ with `float` spelled as the class below:
 * every f32 add / sub / mul / div / sqrt / pow / min / max / floor / compare the oracle executes goes through an overloaded operator
 * that does the same IEEE operation on the same values (results are bit-identical: tests/test_oracle_goldens.py runs a frame
 * through both builds) and counts it.  Built single-threaded (no -fopenmp: the pragmas are ignored), so the counters are plain.
 * Classes (r3o_tally[]): 0 add/sub, 1 mul, 2 div, 3 sqrt, 4 pow / log2 (transcendental), 5 fma (fmaf: one op, two flops),
 * 6 min / max / floor / ceil / rint / abs / negate / compare (vector instructions, not flops), 7 int <-> float conversions.
 * Per SCOPE (r3o.c R3O_SCOPE): the oracle's resolve redoes the triangle setup and the vertex stage for every pixel, the reference
 * runs vs_main per vertex and fs_main per fragment, so the counts are kept apart: 0 fixed-function work, 1 vs_main, 2 fs_main.
 * bench.py quotes flops = [0] + [1] + [2] + [3] + [4] + 2 x [5] of fs_main per shaded pixel (and of vs_main per vertex) beside the
 * counter figure of the HIP kernel. */
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

extern "C" unsigned long long r3o_tally_all[3][8];
extern "C" unsigned long long *r3o_tally;  /* the counters of the current scope (r3o.c R3O_SCOPE: 0 fixed function, 1 vs_main, 2 fs_main) */
#define R3O_SCOPE(n) ((void)(r3o_tally = r3o_tally_all[n]))

struct r3o_tf {
    float v;
    r3o_tf() = default;
    r3o_tf(float x) : v(x) {}
    r3o_tf(double x) : v((float)x) {}
    r3o_tf(int x) : v((float)x) { ++r3o_tally[7]; }
    r3o_tf(unsigned x) : v((float)x) { ++r3o_tally[7]; }
    r3o_tf(long x) : v((float)x) { ++r3o_tally[7]; }
    r3o_tf(unsigned long x) : v((float)x) { ++r3o_tally[7]; }
    r3o_tf(long long x) : v((float)x) { ++r3o_tally[7]; }
    r3o_tf(unsigned long long x) : v((float)x) { ++r3o_tally[7]; }
    explicit operator double() const { return (double)v; }
    explicit operator int() const { ++r3o_tally[7]; return (int)v; }
    explicit operator unsigned() const { ++r3o_tally[7]; return (unsigned)v; }
    explicit operator long() const { ++r3o_tally[7]; return (long)v; }
    explicit operator unsigned long() const { ++r3o_tally[7]; return (unsigned long)v; }
    explicit operator long long() const { ++r3o_tally[7]; return (long long)v; }
    explicit operator unsigned long long() const { ++r3o_tally[7]; return (unsigned long long)v; }
    explicit operator unsigned short() const { ++r3o_tally[7]; return (unsigned short)v; }
    explicit operator unsigned char() const { ++r3o_tally[7]; return (unsigned char)v; }
    explicit operator bool() const { return v != 0.0f; }
    r3o_tf &operator+=(r3o_tf o) { ++r3o_tally[0]; v = v + o.v; return *this; }
    r3o_tf &operator-=(r3o_tf o) { ++r3o_tally[0]; v = v - o.v; return *this; }
    r3o_tf &operator*=(r3o_tf o) { ++r3o_tally[1]; v = v * o.v; return *this; }
    r3o_tf &operator/=(r3o_tf o) { ++r3o_tally[2]; v = v / o.v; return *this; }
};
static_assert(sizeof(r3o_tf) == 4, "same layout as float: the buffers the callers hand over are float arrays");
inline r3o_tf operator+(r3o_tf a, r3o_tf b) { ++r3o_tally[0]; return r3o_tf(a.v + b.v); }
inline r3o_tf operator-(r3o_tf a, r3o_tf b) { ++r3o_tally[0]; return r3o_tf(a.v - b.v); }
inline r3o_tf operator*(r3o_tf a, r3o_tf b) { ++r3o_tally[1]; return r3o_tf(a.v * b.v); }
inline r3o_tf operator/(r3o_tf a, r3o_tf b) { ++r3o_tally[2]; return r3o_tf(a.v / b.v); }
inline r3o_tf operator-(r3o_tf a) { ++r3o_tally[6]; return r3o_tf(-a.v); }
inline r3o_tf operator+(r3o_tf a) { return a; }
#define R3O_TF_CMP(op) inline bool operator op(r3o_tf a, r3o_tf b) { ++r3o_tally[6]; return a.v op b.v; }
R3O_TF_CMP(<) R3O_TF_CMP(<=) R3O_TF_CMP(>) R3O_TF_CMP(>=) R3O_TF_CMP(==) R3O_TF_CMP(!=)
#undef R3O_TF_CMP
inline bool operator!(r3o_tf a) { return a.v == 0.0f; }
inline r3o_tf sqrtf(r3o_tf a) { ++r3o_tally[3]; return r3o_tf(::sqrtf(a.v)); }
inline r3o_tf powf(r3o_tf a, r3o_tf b) { ++r3o_tally[4]; return r3o_tf(::powf(a.v, b.v)); }
inline r3o_tf fmaf(r3o_tf a, r3o_tf b, r3o_tf c) { ++r3o_tally[5]; return r3o_tf(::fmaf(a.v, b.v, c.v)); }
inline r3o_tf fminf(r3o_tf a, r3o_tf b) { ++r3o_tally[6]; return r3o_tf(::fminf(a.v, b.v)); }
inline r3o_tf fmaxf(r3o_tf a, r3o_tf b) { ++r3o_tally[6]; return r3o_tf(::fmaxf(a.v, b.v)); }
inline r3o_tf floorf(r3o_tf a) { ++r3o_tally[6]; return r3o_tf(::floorf(a.v)); }
inline r3o_tf ceilf(r3o_tf a) { ++r3o_tally[6]; return r3o_tf(::ceilf(a.v)); }
inline r3o_tf rintf(r3o_tf a) { ++r3o_tally[6]; return r3o_tf(::rintf(a.v)); }
inline r3o_tf fabsf(r3o_tf a) { ++r3o_tally[6]; return r3o_tf(::fabsf(a.v)); }
inline r3o_tf frexpf(r3o_tf a, int *e) { ++r3o_tally[6]; return r3o_tf(::frexpf(a.v, e)); }
inline bool r3o_tf_isinf(r3o_tf a) { return isinf(a.v); }
/* the double-precision helpers the oracle uses on f32 inputs (level selection, sRGB tables): forwarded, counted as transcendental */
inline double log2(r3o_tf a) { ++r3o_tally[4]; return ::log2((double)a.v); }
inline double sqrt(r3o_tf a) { ++r3o_tally[3]; return ::sqrt((double)a.v); }
inline double floor(r3o_tf a) { ++r3o_tally[6]; return ::floor((double)a.v); }
inline double pow(r3o_tf a, double b) { ++r3o_tally[4]; return ::pow((double)a.v, b); }
#undef isinf
#define isinf(x) r3o_tf_isinf(x)
#define float r3o_tf
