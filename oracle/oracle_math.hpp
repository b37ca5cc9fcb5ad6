// TEST INFRASTRUCTURE — CPU restatement (the parity checker), not the product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/.
//
// Small f32 linear algebra with a FIXED operation order (compiled with -ffp-contract=off), so
// that every value is reproducible bit for bit.  The formulas restate the published algorithms
// of the reference's third-party math crate glam ("0.24", reference Cargo.toml; not vendored
// under /root/reference): Mat4::perspective_lh (used at src/camera.rs:27-29),
// Mat4::orthographic_lh / Mat4::look_to_lh (src/renderer/visual_server.rs:826,:855),
// Affine3A::inverse (visual_server.rs:208), Mat4::inverse (visual_server.rs:210).
// PARITY UNPINNED: the reference holds no golden vectors for any of them.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };
// column-major: c[col*4 + row], as glam Mat4::to_cols_array()
struct M4 { float c[16]; };

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
// dot = (x*x' + y*y') + z*z'
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// WGSL normalize(v) = v / length(v); stated here as v * (1/sqrt(dot(v,v)))
inline V3 normalize(V3 a) {
    float inv = 1.0f / std::sqrt(dot(a, a));
    return a * inv;
}

inline M4 identity() {
    M4 m{};
    m.c[0] = m.c[5] = m.c[10] = m.c[15] = 1.0f;
    return m;
}

// M * v, summed left to right over columns: ((c0*x + c1*y) + c2*z) + c3*w
inline V4 mul(const M4& m, V4 v) {
    V4 r;
    r.x = ((m.c[0] * v.x + m.c[4] * v.y) + m.c[8] * v.z) + m.c[12] * v.w;
    r.y = ((m.c[1] * v.x + m.c[5] * v.y) + m.c[9] * v.z) + m.c[13] * v.w;
    r.z = ((m.c[2] * v.x + m.c[6] * v.y) + m.c[10] * v.z) + m.c[14] * v.w;
    r.w = ((m.c[3] * v.x + m.c[7] * v.y) + m.c[11] * v.z) + m.c[15] * v.w;
    return r;
}
// point (w = 1): the last term is the column itself
inline V4 mul_point(const M4& m, V3 p) {
    V4 r;
    r.x = ((m.c[0] * p.x + m.c[4] * p.y) + m.c[8] * p.z) + m.c[12];
    r.y = ((m.c[1] * p.x + m.c[5] * p.y) + m.c[9] * p.z) + m.c[13];
    r.z = ((m.c[2] * p.x + m.c[6] * p.y) + m.c[10] * p.z) + m.c[14];
    r.w = ((m.c[3] * p.x + m.c[7] * p.y) + m.c[11] * p.z) + m.c[15];
    return r;
}
// direction (w = 0): xyz only
inline V3 mul_dir(const M4& m, V3 d) {
    V3 r;
    r.x = (m.c[0] * d.x + m.c[4] * d.y) + m.c[8] * d.z;
    r.y = (m.c[1] * d.x + m.c[5] * d.y) + m.c[9] * d.z;
    r.z = (m.c[2] * d.x + m.c[6] * d.y) + m.c[10] * d.z;
    return r;
}
// A * B, column j of the product = A * (column j of B)
inline M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int j = 0; j < 4; ++j) {
        V4 col = mul(a, V4{b.c[j * 4 + 0], b.c[j * 4 + 1], b.c[j * 4 + 2], b.c[j * 4 + 3]});
        r.c[j * 4 + 0] = col.x;
        r.c[j * 4 + 1] = col.y;
        r.c[j * 4 + 2] = col.z;
        r.c[j * 4 + 3] = col.w;
    }
    return r;
}

// glam Mat4::perspective_lh(fov_y, aspect, near, far): depth 0..1, +Z forward.
inline M4 perspective_lh(float fov_y, float aspect, float z_near, float z_far) {
    float s = std::sin(0.5f * fov_y), c = std::cos(0.5f * fov_y);
    float h = c / s;
    float w = h / aspect;
    float r = z_far / (z_far - z_near);
    M4 m{};
    m.c[0] = w;
    m.c[5] = h;
    m.c[10] = r;
    m.c[11] = 1.0f;
    m.c[14] = -r * z_near;
    return m;
}

// glam Mat4::orthographic_lh(left, right, bottom, top, near, far)
inline M4 orthographic_lh(float l, float r, float b, float t, float n, float f) {
    float rcp_w = 1.0f / (r - l), rcp_h = 1.0f / (t - b), rr = 1.0f / (f - n);
    M4 m{};
    m.c[0] = rcp_w + rcp_w;
    m.c[5] = rcp_h + rcp_h;
    m.c[10] = rr;
    m.c[12] = -(l + r) * rcp_w;
    m.c[13] = -(t + b) * rcp_h;
    m.c[14] = -rr * n;
    m.c[15] = 1.0f;
    return m;
}

// glam Mat4::look_to_lh(eye, dir, up): f = normalize(dir), s = normalize(up x f), u = f x s
inline M4 look_to_lh(V3 eye, V3 dir, V3 up) {
    V3 f = normalize(dir);
    V3 s = normalize(cross(up, f));
    V3 u = cross(f, s);
    M4 m{};
    m.c[0] = s.x; m.c[1] = u.x; m.c[2] = f.x;  m.c[3] = 0.0f;
    m.c[4] = s.y; m.c[5] = u.y; m.c[6] = f.y;  m.c[7] = 0.0f;
    m.c[8] = s.z; m.c[9] = u.z; m.c[10] = f.z; m.c[11] = 0.0f;
    m.c[12] = -dot(s, eye); m.c[13] = -dot(u, eye); m.c[14] = -dot(f, eye); m.c[15] = 1.0f;
    return m;
}

// glam Affine3A::inverse: 3x3 inverse by cross products / determinant, then t' = -(inv3 * t).
// Input and output as Mat4 with last row (0,0,0,1).
inline M4 affine_inverse(const M4& a) {
    V3 x{a.c[0], a.c[1], a.c[2]}, y{a.c[4], a.c[5], a.c[6]}, z{a.c[8], a.c[9], a.c[10]};
    V3 t{a.c[12], a.c[13], a.c[14]};
    V3 t0 = cross(y, z), t1 = cross(z, x), t2 = cross(x, y);
    float det = dot(z, t2);
    float inv_det = 1.0f / det;
    // rows of the inverse are t0,t1,t2 scaled; stored column-major => transpose
    V3 r0 = t0 * inv_det, r1 = t1 * inv_det, r2 = t2 * inv_det;
    M4 m{};
    m.c[0] = r0.x; m.c[1] = r1.x; m.c[2] = r2.x;
    m.c[4] = r0.y; m.c[5] = r1.y; m.c[6] = r2.y;
    m.c[8] = r0.z; m.c[9] = r1.z; m.c[10] = r2.z;
    V3 nt = mul_dir(m, t);
    m.c[12] = -nt.x; m.c[13] = -nt.y; m.c[14] = -nt.z;
    m.c[15] = 1.0f;
    return m;
}

// General 4x4 inverse by cofactors (glam Mat4::inverse computes the same adjugate / det;
// summation order here is this file's own and is what the product must match).
inline M4 inverse(const M4& m) {
    const float* a = m.c;
    float a00 = a[0], a01 = a[1], a02 = a[2], a03 = a[3];
    float a10 = a[4], a11 = a[5], a12 = a[6], a13 = a[7];
    float a20 = a[8], a21 = a[9], a22 = a[10], a23 = a[11];
    float a30 = a[12], a31 = a[13], a32 = a[14], a33 = a[15];
    float b00 = a00 * a11 - a01 * a10;
    float b01 = a00 * a12 - a02 * a10;
    float b02 = a00 * a13 - a03 * a10;
    float b03 = a01 * a12 - a02 * a11;
    float b04 = a01 * a13 - a03 * a11;
    float b05 = a02 * a13 - a03 * a12;
    float b06 = a20 * a31 - a21 * a30;
    float b07 = a20 * a32 - a22 * a30;
    float b08 = a20 * a33 - a23 * a30;
    float b09 = a21 * a32 - a22 * a31;
    float b10 = a21 * a33 - a23 * a31;
    float b11 = a22 * a33 - a23 * a32;
    float det = ((((b00 * b11 - b01 * b10) + b02 * b09) + b03 * b08) - b04 * b07) + b05 * b06;
    float id = 1.0f / det;
    M4 r;
    r.c[0] = ((a11 * b11 - a12 * b10) + a13 * b09) * id;
    r.c[1] = ((a02 * b10 - a01 * b11) - a03 * b09) * id;
    r.c[2] = ((a31 * b05 - a32 * b04) + a33 * b03) * id;
    r.c[3] = ((a22 * b04 - a21 * b05) - a23 * b03) * id;
    r.c[4] = ((a12 * b08 - a10 * b11) - a13 * b07) * id;
    r.c[5] = ((a00 * b11 - a02 * b08) + a03 * b07) * id;
    r.c[6] = ((a32 * b02 - a30 * b05) - a33 * b01) * id;
    r.c[7] = ((a20 * b05 - a22 * b02) + a23 * b01) * id;
    r.c[8] = ((a10 * b10 - a11 * b08) + a13 * b06) * id;
    r.c[9] = ((a01 * b08 - a00 * b10) - a03 * b06) * id;
    r.c[10] = ((a30 * b04 - a31 * b02) + a33 * b00) * id;
    r.c[11] = ((a21 * b02 - a20 * b04) - a23 * b00) * id;
    r.c[12] = ((a11 * b07 - a10 * b09) - a12 * b06) * id;
    r.c[13] = ((a00 * b09 - a01 * b07) + a02 * b06) * id;
    r.c[14] = ((a31 * b01 - a30 * b03) - a32 * b00) * id;
    r.c[15] = ((a20 * b03 - a21 * b01) + a22 * b00) * id;
    return r;
}

// IEEE binary16 round-to-nearest-even of an f32, returned widened back to f32
// (the Rgba16Float colour target, visual_server.rs:92-100).
inline float round_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    uint32_t sign = x & 0x80000000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return f;                 // inf / nan
    if (ax >= 0x477FF000u) {                          // >= 65520 rounds to inf
        uint32_t r = sign | 0x7F800000u;
        float o; std::memcpy(&o, &r, 4); return o;
    }
    if (ax < 0x38800000u) {                           // below 2^-14: f16 subnormal, spacing 2^-24
        float a; std::memcpy(&a, &ax, 4);
        // adding 2^-1 forces rounding to a multiple of 2^-24 (ulp of [0.5,1) in f32), RNE
        float r = (a + 0.5f) - 0.5f;
        uint32_t rb; std::memcpy(&rb, &r, 4); rb |= sign;
        float o; std::memcpy(&o, &rb, 4); return o;
    }
    // normal: keep 10 mantissa bits, RNE on the 13 dropped bits
    uint32_t lsb = (ax >> 13) & 1u;
    ax += 0x0FFFu + lsb;
    ax &= 0xFFFFE000u;
    uint32_t r = sign | ax;
    float o; std::memcpy(&o, &r, 4); return o;
}

inline float clampf(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }

// sRGB OETF + unorm8 conversion of the sRGB surface (backend.rs:75-80).
inline uint8_t srgb8(float lin) {
    float x = clampf(lin, 0.0f, 1.0f);
    float y = (x <= 0.0031308f) ? 12.92f * x : 1.055f * std::pow(x, 1.0f / 2.4f) - 0.055f;
    float q = y * 255.0f + 0.5f;
    int v = (int)q;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

}  // namespace orc
