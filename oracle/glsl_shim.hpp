// oracle/glsl_shim.hpp — TEST INFRASTRUCTURE.  Just enough of GLSL ES 3.00 in C++ for the reference's splat shaders to
// compile AS THEY ARE: oracle/make_golden_raster.py asks the reference's own SplatMaterial3D.build() for its vertex and
// fragment shader strings (oracle/shader_dump.mjs), applies a few mechanical token rewrites (listed there: qualifiers
// dropped, float literals suffixed, `float[](...)` -> `{...}`, `discard` -> a macro) and compiles the result against this
// header with g++ -O1 -ffp-contract=off: every arithmetic statement of the shaders then runs in IEEE fp32, in the order the
// shader text states it.  Vector types carry GLSL swizzles as proxy members of an anonymous union (storage shared with the
// components), operators are plain overloads so that proxies convert implicitly.  Nothing here comes from the reference.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

namespace glsl {

typedef unsigned int uint;
struct vec2; struct vec3; struct vec4; struct uvec3; struct uvec4;

// swizzle proxies: same storage as the vector's components
template <class V, class T, int A, int B> struct Sw2 {
    T d[4];
    operator V() const;
    Sw2& operator=(const V& v);
};
template <class V, class T, int A, int B, int C> struct Sw3 {
    T d[4];
    operator V() const;
    Sw3& operator=(const V& v);
    Sw3& operator+=(const V& v);
    Sw3& operator*=(float s);
};

struct vec2 {
    union {
        struct { float x, y; };
        struct { float r, g; };
        float d[2];
        Sw2<vec2, float, 0, 1> xy, rg;
    };
    vec2() : x(0), y(0) {}
    explicit vec2(float s) : x(s), y(s) {}
    vec2(float a, float b) : x(a), y(b) {}
    vec2(const vec2& o) : x(o.x), y(o.y) {}
    vec2& operator=(const vec2& o) { x = o.x; y = o.y; return *this; }
    vec2& operator*=(float s) { x *= s; y *= s; return *this; }
    float& operator[](int i) { return d[i]; }
};
struct vec3 {
    union {
        struct { float x, y, z; };
        struct { float r, g, b; };
        float d[3];
        Sw2<vec2, float, 0, 1> xy, rg;
        Sw3<vec3, float, 0, 1, 2> xyz, rgb;
    };
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    vec3(const vec2& a, float c) : x(a.x), y(a.y), z(c) {}
    vec3(float a, const vec2& b) : x(a), y(b.x), z(b.y) {}
    vec3(const vec3& o) : x(o.x), y(o.y), z(o.z) {}
    explicit vec3(const vec4& v);
    vec3& operator=(const vec3& o) { x = o.x; y = o.y; z = o.z; return *this; }
    float& operator[](int i) { return d[i]; }
    const float& operator[](int i) const { return d[i]; }
};
struct vec4 {
    union {
        struct { float x, y, z, w; };
        struct { float r, g, b, a; };
        float d[4];
        Sw2<vec2, float, 0, 1> xy, rg;
        Sw2<vec2, float, 2, 3> zw, ba;
        Sw3<vec3, float, 0, 1, 2> xyz, rgb;
        Sw3<vec3, float, 1, 2, 3> yzw, gba;
    };
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a, float b, float c, float e) : x(a), y(b), z(c), w(e) {}
    vec4(const vec3& v, float e) : x(v.x), y(v.y), z(v.z), w(e) {}
    vec4(const vec2& v, float c, float e) : x(v.x), y(v.y), z(c), w(e) {}
    vec4(const vec4& o) : x(o.x), y(o.y), z(o.z), w(o.w) {}
    explicit vec4(const uvec4& u);
    vec4& operator=(const vec4& o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
    float& operator[](int i) { return d[i]; }
    const float& operator[](int i) const { return d[i]; }
};
struct uvec3 {
    union { struct { uint x, y, z; }; struct { uint r, g, b; }; uint d[3]; };
    uvec3() : x(0), y(0), z(0) {}
    uvec3(uint a, uint b, uint c) : x(a), y(b), z(c) {}
    uvec3(const uvec3& o) : x(o.x), y(o.y), z(o.z) {}
    uvec3& operator=(const uvec3& o) { x = o.x; y = o.y; z = o.z; return *this; }
};
struct uvec4 {
    union {
        struct { uint x, y, z, w; };
        struct { uint r, g, b, a; };
        uint d[4];
        Sw3<uvec3, uint, 1, 2, 3> yzw, gba;
    };
    uvec4() : x(0), y(0), z(0), w(0) {}
    uvec4(uint a, uint b, uint c, uint e) : x(a), y(b), z(c), w(e) {}
    uvec4(const uvec4& o) : x(o.x), y(o.y), z(o.z), w(o.w) {}
    uvec4& operator=(const uvec4& o) { x = o.x; y = o.y; z = o.z; w = o.w; return *this; }
};
inline vec3::vec3(const vec4& v) : x(v.x), y(v.y), z(v.z) {}
inline vec4::vec4(const uvec4& u) : x((float)u.x), y((float)u.y), z((float)u.z), w((float)u.w) {}

template <class V, class T, int A, int B> Sw2<V, T, A, B>::operator V() const { return V(d[A], d[B]); }
template <class V, class T, int A, int B> Sw2<V, T, A, B>& Sw2<V, T, A, B>::operator=(const V& v) {
    const T a = v.d[0], b = v.d[1];
    d[A] = a; d[B] = b;
    return *this;
}
template <class V, class T, int A, int B, int C> Sw3<V, T, A, B, C>::operator V() const { return V(d[A], d[B], d[C]); }
template <class V, class T, int A, int B, int C> Sw3<V, T, A, B, C>& Sw3<V, T, A, B, C>::operator=(const V& v) {
    const T a = v.d[0], b = v.d[1], c = v.d[2];
    d[A] = a; d[B] = b; d[C] = c;
    return *this;
}
template <class V, class T, int A, int B, int C> Sw3<V, T, A, B, C>& Sw3<V, T, A, B, C>::operator+=(const V& v) {
    d[A] = d[A] + v.d[0]; d[B] = d[B] + v.d[1]; d[C] = d[C] + v.d[2];
    return *this;
}
template <class V, class T, int A, int B, int C> Sw3<V, T, A, B, C>& Sw3<V, T, A, B, C>::operator*=(float s) {
    d[A] = d[A] * s; d[B] = d[B] * s; d[C] = d[C] * s;
    return *this;
}

// component-wise arithmetic (plain overloads: swizzle proxies convert implicitly)
#define GLSL_VEC_OPS(V, N)                                                                                   \
    inline V operator+(const V& a, const V& b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i]; return r; } \
    inline V operator-(const V& a, const V& b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i]; return r; } \
    inline V operator*(const V& a, const V& b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.d[i]; return r; } \
    inline V operator/(const V& a, const V& b) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] / b.d[i]; return r; } \
    inline V operator*(const V& a, float s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] * s; return r; }         \
    inline V operator*(float s, const V& a) { V r; for (int i = 0; i < N; i++) r.d[i] = s * a.d[i]; return r; }         \
    inline V operator/(const V& a, float s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] / s; return r; }         \
    inline V operator+(const V& a, float s) { V r; for (int i = 0; i < N; i++) r.d[i] = a.d[i] + s; return r; }         \
    inline V operator-(const V& a) { V r; for (int i = 0; i < N; i++) r.d[i] = -a.d[i]; return r; }                     \
    inline float dot(const V& a, const V& b) { float s = a.d[0] * b.d[0]; for (int i = 1; i < N; i++) s = s + a.d[i] * b.d[i]; return s; } \
    inline float length(const V& a) { return sqrtf(dot(a, a)); }                                             \
    inline V normalize(const V& a) { return a / length(a); }                                                 \
    inline V clamp(const V& v, const V& lo, const V& hi) { V r; for (int i = 0; i < N; i++) r.d[i] = fminf(fmaxf(v.d[i], lo.d[i]), hi.d[i]); return r; }
GLSL_VEC_OPS(vec2, 2)
GLSL_VEC_OPS(vec3, 3)
GLSL_VEC_OPS(vec4, 4)

inline uvec4 operator&(const uvec4& a, uint u) { return uvec4(a.x & u, a.y & u, a.z & u, a.w & u); }
inline uvec4 operator>>(const uvec4& a, const uvec4& s) { return uvec4(a.x >> s.x, a.y >> s.y, a.z >> s.z, a.w >> s.w); }

// scalar built-ins in fp32 (sqrt / exp / floor of a float resolve to <cmath>'s float overloads)
inline float fract(float v) { return v - floorf(v); }
inline float max(float a, float b) { return a < b ? b : a; }      // GLSL: y if x < y
inline float min(float a, float b) { return b < a ? b : a; }
inline float clamp(float v, float lo, float hi) { return min(max(v, lo), hi); }
inline float step(float edge, float v) { return v < edge ? 0.0f : 1.0f; }
inline vec3 uintBitsToFloat(const uvec3& u) {
    vec3 r;
    memcpy(r.d, u.d, 12);
    return r;
}
inline float half_to_float(uint h) {
    const uint sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    uint bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else {
            int s = 0;
            uint mm = m;
            while (!(mm & 1024u)) { mm <<= 1; s++; }
            bits = sign | ((uint)(113 - s) << 23) | ((mm & 1023u) << 13);
        }
    } else if (e == 31) bits = sign | 0x7F800000u | (m << 13);
    else bits = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
inline vec2 unpackHalf2x16(uint v) { return vec2(half_to_float(v & 0xFFFFu), half_to_float(v >> 16)); }

// matrices: column-major, m[col][row]
struct mat4;
struct mat3 {
    vec3 c[3];
    mat3() {}
    mat3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
        c[0] = vec3(a0, a1, a2); c[1] = vec3(b0, b1, b2); c[2] = vec3(c0, c1, c2);
    }
    explicit mat3(const mat4& m);
    vec3& operator[](int i) { return c[i]; }
    const vec3& operator[](int i) const { return c[i]; }
};
struct mat4 {
    vec4 c[4];
    mat4() {}
    vec4& operator[](int i) { return c[i]; }
    const vec4& operator[](int i) const { return c[i]; }
};
inline mat3::mat3(const mat4& m) { for (int j = 0; j < 3; j++) c[j] = vec3(m[j].x, m[j].y, m[j].z); }
inline mat3 transpose(const mat3& m) {
    mat3 r;
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) r[j][i] = m[i][j];
    return r;
}
inline mat3 operator*(const mat3& a, const mat3& b) {            // (a*b)[j][i] = sum_k a[k][i] * b[j][k]
    mat3 r;
    for (int j = 0; j < 3; j++) for (int i = 0; i < 3; i++) r[j][i] = a[0][i] * b[j][0] + a[1][i] * b[j][1] + a[2][i] * b[j][2];
    return r;
}
inline mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r;
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) r[j][i] = a[0][i] * b[j][0] + a[1][i] * b[j][1] + a[2][i] * b[j][2] + a[3][i] * b[j][3];
    return r;
}
inline vec4 operator*(const mat4& m, const vec4& v) {
    vec4 r;
    for (int i = 0; i < 4; i++) r[i] = m[0][i] * v.x + m[1][i] * v.y + m[2][i] * v.z + m[3][i] * v.w;
    return r;
}
inline mat4 inverse(const mat4& m) {                             // cofactor expansion in fp32 (GLSL leaves the precision open)
    float a[16], inv[16];
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) a[4 * j + i] = m[j][i];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    mat4 r;
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) r[j][i] = inv[4 * j + i] / det;
    return r;
}

// textures: NEAREST, texel = floor(uv * size) (the harness uses power-of-two sizes, so the shaders' index -> uv -> texel
// round trip is exact)
struct sampler2D {
    const float* data = nullptr;       // RGBA float texels (half / unorm8 storage is converted when the harness fills it)
    int w = 1, h = 1;
};
struct usampler2D {
    const uint* data = nullptr;        // channels per texel below
    int w = 1, h = 1, channels = 4;
};
inline int texel_index(const vec2& uv, int w, int h) {
    int ix = (int)floorf(uv.x * (float)w), iy = (int)floorf(uv.y * (float)h);
    ix = ix < 0 ? 0 : (ix >= w ? w - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= h ? h - 1 : iy);
    return iy * w + ix;
}
inline vec4 texture(const sampler2D& s, const vec2& uv) {
    const float* t = s.data + 4 * (size_t)texel_index(uv, s.w, s.h);
    return vec4(t[0], t[1], t[2], t[3]);
}
inline uvec4 texture(const usampler2D& s, const vec2& uv) {
    const uint* t = s.data + (size_t)s.channels * texel_index(uv, s.w, s.h);
    return uvec4(t[0], s.channels > 1 ? t[1] : 0u, s.channels > 2 ? t[2] : 0u, s.channels > 3 ? t[3] : 1u);
}

}  // namespace glsl
