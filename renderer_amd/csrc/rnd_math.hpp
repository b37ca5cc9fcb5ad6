// Host-side f32 math of the frame driver: the glam calls the reference makes on the hot path
// (Mat4::perspective_lh at src/camera.rs:27-29; Affine3A::inverse, Mat4 * Mat4 and Mat4::inverse at
// src/renderer/visual_server.rs:206-214; look_to_lh / orthographic_lh at :826, :855).
// Operation order is fixed (see DESIGN.md "Host math"); build with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace rnd {

struct Vec3 {
    float x = 0, y = 0, z = 0;
};
struct Vec4 {
    float x = 0, y = 0, z = 0, w = 0;
};

inline Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 scale(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 normalized(Vec3 a) { return scale(a, 1.0f / std::sqrt(dot(a, a))); }

// Column-major 4x4: col[j] is column j, exactly glam's Mat4 { x_axis, y_axis, z_axis, w_axis }.
struct Mat4 {
    Vec4 col[4];

    static Mat4 identity() {
        Mat4 m;
        m.col[0].x = m.col[1].y = m.col[2].z = m.col[3].w = 1.0f;
        return m;
    }
    static Mat4 from_cols_array(const float* a) {
        Mat4 m;
        std::memcpy(&m, a, sizeof(float) * 16);
        return m;
    }
    void to_cols_array(float* out) const { std::memcpy(out, this, sizeof(float) * 16); }
    Vec3 axis(int j) const { return {col[j].x, col[j].y, col[j].z}; }
};
static_assert(sizeof(Mat4) == 64, "Mat4 must be 16 packed floats");

// column sum, left to right: ((x_axis*v.x + y_axis*v.y) + z_axis*v.z) + w_axis*v.w
inline Vec4 transform(const Mat4& m, Vec4 v) {
    Vec4 r;
    r.x = ((m.col[0].x * v.x + m.col[1].x * v.y) + m.col[2].x * v.z) + m.col[3].x * v.w;
    r.y = ((m.col[0].y * v.x + m.col[1].y * v.y) + m.col[2].y * v.z) + m.col[3].y * v.w;
    r.z = ((m.col[0].z * v.x + m.col[1].z * v.y) + m.col[2].z * v.z) + m.col[3].z * v.w;
    r.w = ((m.col[0].w * v.x + m.col[1].w * v.y) + m.col[2].w * v.z) + m.col[3].w * v.w;
    return r;
}
inline Vec3 transform_dir(const Mat4& m, Vec3 d) {
    return {(m.col[0].x * d.x + m.col[1].x * d.y) + m.col[2].x * d.z,
            (m.col[0].y * d.x + m.col[1].y * d.y) + m.col[2].y * d.z,
            (m.col[0].z * d.x + m.col[1].z * d.y) + m.col[2].z * d.z};
}
inline Mat4 operator*(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int j = 0; j < 4; ++j) r.col[j] = transform(a, b.col[j]);
    return r;
}

inline Mat4 perspective_lh(float fov_y, float aspect, float z_near, float z_far) {
    const float half = 0.5f * fov_y;
    const float sin_fov = std::sin(half), cos_fov = std::cos(half);
    const float h = cos_fov / sin_fov;
    const float w = h / aspect;
    const float r = z_far / (z_far - z_near);
    Mat4 m;
    m.col[0].x = w;
    m.col[1].y = h;
    m.col[2].z = r;
    m.col[2].w = 1.0f;
    m.col[3].z = -r * z_near;
    return m;
}

inline Mat4 orthographic_lh(float left, float right, float bottom, float top, float z_near, float z_far) {
    const float rcp_width = 1.0f / (right - left);
    const float rcp_height = 1.0f / (top - bottom);
    const float r = 1.0f / (z_far - z_near);
    Mat4 m;
    m.col[0].x = rcp_width + rcp_width;
    m.col[1].y = rcp_height + rcp_height;
    m.col[2].z = r;
    m.col[3] = {-(left + right) * rcp_width, -(top + bottom) * rcp_height, -r * z_near, 1.0f};
    return m;
}

inline Mat4 look_to_lh(Vec3 eye, Vec3 dir, Vec3 up) {
    const Vec3 f = normalized(dir);
    const Vec3 s = normalized(cross(up, f));
    const Vec3 u = cross(f, s);
    Mat4 m;
    m.col[0] = {s.x, u.x, f.x, 0.0f};
    m.col[1] = {s.y, u.y, f.y, 0.0f};
    m.col[2] = {s.z, u.z, f.z, 0.0f};
    m.col[3] = {-dot(s, eye), -dot(u, eye), -dot(f, eye), 1.0f};
    return m;
}

// inverse of an affine transform held as Mat4 (last row 0,0,0,1)
inline Mat4 affine_inverse(const Mat4& a) {
    const Vec3 x = a.axis(0), y = a.axis(1), z = a.axis(2), t = a.axis(3);
    const Vec3 c0 = cross(y, z), c1 = cross(z, x), c2 = cross(x, y);
    const float inv_det = 1.0f / dot(z, c2);
    const Vec3 r0 = scale(c0, inv_det), r1 = scale(c1, inv_det), r2 = scale(c2, inv_det);
    Mat4 m;
    m.col[0] = {r0.x, r1.x, r2.x, 0.0f};
    m.col[1] = {r0.y, r1.y, r2.y, 0.0f};
    m.col[2] = {r0.z, r1.z, r2.z, 0.0f};
    const Vec3 nt = transform_dir(m, t);
    m.col[3] = {-nt.x, -nt.y, -nt.z, 1.0f};
    return m;
}

// general inverse: 2x2 sub-determinants, adjugate / determinant
inline Mat4 inverse(const Mat4& mm) {
    float a[16];
    mm.to_cols_array(a);
    const float s0 = a[0] * a[5] - a[1] * a[4], s1 = a[0] * a[6] - a[2] * a[4], s2 = a[0] * a[7] - a[3] * a[4];
    const float s3 = a[1] * a[6] - a[2] * a[5], s4 = a[1] * a[7] - a[3] * a[5], s5 = a[2] * a[7] - a[3] * a[6];
    const float t0 = a[8] * a[13] - a[9] * a[12], t1 = a[8] * a[14] - a[10] * a[12], t2 = a[8] * a[15] - a[11] * a[12];
    const float t3 = a[9] * a[14] - a[10] * a[13], t4 = a[9] * a[15] - a[11] * a[13], t5 = a[10] * a[15] - a[11] * a[14];
    const float det = ((((s0 * t5 - s1 * t4) + s2 * t3) + s3 * t2) - s4 * t1) + s5 * t0;
    const float id = 1.0f / det;
    float r[16];
    r[0] = ((a[5] * t5 - a[6] * t4) + a[7] * t3) * id;
    r[1] = ((a[2] * t4 - a[1] * t5) - a[3] * t3) * id;
    r[2] = ((a[13] * s5 - a[14] * s4) + a[15] * s3) * id;
    r[3] = ((a[10] * s4 - a[9] * s5) - a[11] * s3) * id;
    r[4] = ((a[6] * t2 - a[4] * t5) - a[7] * t1) * id;
    r[5] = ((a[0] * t5 - a[2] * t2) + a[3] * t1) * id;
    r[6] = ((a[14] * s2 - a[12] * s5) - a[15] * s1) * id;
    r[7] = ((a[8] * s5 - a[10] * s2) + a[11] * s1) * id;
    r[8] = ((a[4] * t4 - a[5] * t2) + a[7] * t0) * id;
    r[9] = ((a[1] * t2 - a[0] * t4) - a[3] * t0) * id;
    r[10] = ((a[12] * s4 - a[13] * s2) + a[15] * s0) * id;
    r[11] = ((a[9] * s2 - a[8] * s4) - a[11] * s0) * id;
    r[12] = ((a[5] * t1 - a[4] * t3) - a[6] * t0) * id;
    r[13] = ((a[0] * t3 - a[1] * t1) + a[2] * t0) * id;
    r[14] = ((a[13] * s1 - a[12] * s3) - a[14] * s0) * id;
    r[15] = ((a[8] * s3 - a[9] * s1) + a[10] * s0) * id;
    return Mat4::from_cols_array(r);
}

// compute_shadow_cascade_projviews (visual_server.rs:771-863): world -> light clip of the four
// shadow cascades of a directional light, from the inverse projection*view current at the call.
// Frustum slices at the ratios of settings.shadow_cascades (visual_server.rs:88).
inline void shadow_cascade_projviews(const Mat4& inv_projection_view, Vec3 light_dir, Mat4 out[4]) {
    auto unproject = [&](float x, float y, float z) {
        const Vec4 p = transform(inv_projection_view, Vec4{x, y, z, 1.0f});
        return Vec4{p.x / p.w, p.y / p.w, p.z / p.w, p.w / p.w};
    };
    const float sx[4] = {-1.0f, 1.0f, -1.0f, 1.0f}, sy[4] = {1.0f, 1.0f, -1.0f, -1.0f};   // tl, tr, bl, br
    Vec4 near_c[4], edge[4];
    for (int k = 0; k < 4; ++k) {
        near_c[k] = unproject(sx[k], sy[k], 0.0f);
        const Vec4 far_c = unproject(sx[k], sy[k], 1.0f);
        edge[k] = Vec4{far_c.x - near_c[k].x, far_c.y - near_c[k].y, far_c.z - near_c[k].z, far_c.w - near_c[k].w};
    }
    static const float kSlices[4][2] = {{0.0f, 0.05f}, {0.05f, 0.1f}, {0.1f, 0.3f}, {0.3f, 1.0f}};
    for (int c = 0; c < 4; ++c) {
        Vec4 corners[8];
        for (int half = 0; half < 2; ++half)
            for (int k = 0; k < 4; ++k) {
                const float r = kSlices[c][half];
                corners[half * 4 + k] = Vec4{near_c[k].x + edge[k].x * r, near_c[k].y + edge[k].y * r,
                                             near_c[k].z + edge[k].z * r, near_c[k].w + edge[k].w * r};
            }
        Vec4 acc = corners[0];
        for (int k = 1; k < 8; ++k) acc = Vec4{acc.x + corners[k].x, acc.y + corners[k].y, acc.z + corners[k].z, acc.w + corners[k].w};
        const Mat4 view = look_to_lh(Vec3{acc.x / 8.0f, acc.y / 8.0f, acc.z / 8.0f}, light_dir, Vec3{0.0f, 1.0f, 0.0f});
        float lo[3] = {3.40282347e+38f, 3.40282347e+38f, 3.40282347e+38f};
        float hi[3] = {-3.40282347e+38f, -3.40282347e+38f, -3.40282347e+38f};
        for (const Vec4& corner : corners) {
            const Vec4 p = transform(view, corner);
            lo[0] = std::fmin(lo[0], p.x); hi[0] = std::fmax(hi[0], p.x);
            lo[1] = std::fmin(lo[1], p.y); hi[1] = std::fmax(hi[1], p.y);
            lo[2] = std::fmin(lo[2], p.z); hi[2] = std::fmax(hi[2], p.z);
        }
        // include casters outside the slice: widen the z range tenfold
        lo[2] = lo[2] < 0.0f ? lo[2] * 10.0f : lo[2] / 10.0f;
        hi[2] = hi[2] < 0.0f ? hi[2] / 10.0f : hi[2] * 10.0f;
        out[c] = orthographic_lh(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]) * view;
    }
}

// IEEE binary16 round-to-nearest-even, widened back to f32 (the Rgba16Float clear colour)
inline float round_to_f16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t mag = u & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return f;                      // inf, nan
    const int exp = (int)(mag >> 23) - 127;
    if (exp > 15 || (exp == 15 && (mag & 0x007FFFFFu) >= 0x007FF000u)) mag = 0x7F800000u;   // >= 65520 -> inf
    else {
        // number of low mantissa bits a binary16 cannot hold: 13 for normals, more below 2^-14
        const int drop = exp >= -14 ? 13 : (exp >= -25 ? 13 + (-14 - exp) : 25);
        if (drop >= 25) mag = 0;                           // below half the smallest subnormal
        else {
            const uint32_t full = (mag & 0x007FFFFFu) | 0x00800000u;   // explicit leading one
            const uint32_t unit = 1u << drop, rem = full & (unit - 1u), half = unit >> 1;
            uint32_t kept = full >> drop;
            if (rem > half || (rem == half && (kept & 1u))) kept += 1;
            // rebuild: value = kept * 2^(exp - 23 + drop)
            float v = std::ldexp((float)kept, exp - 23 + drop);
            std::memcpy(&mag, &v, 4);
        }
    }
    const uint32_t out = sign | mag;
    float r;
    std::memcpy(&r, &out, 4);
    return r;
}

}  // namespace rnd
