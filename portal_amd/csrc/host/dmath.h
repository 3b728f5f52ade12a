// dmath.h -- binary64 vector / quaternion / matrix arithmetic for scene constants.
//
// Replaces the third-party `glam 0.13.1` crate (Cargo.lock:718) at the reference call sites
// src/gui/matrix.rs:537-547,555-569,619-629 (from_scale_rotation_translation, DQuat
// rotations, inverse), src/gui/scene.rs:587-588,626,631 (inverse, mul, as_f32) and
// src/main.rs:286-304 (camera basis).  glam is not vendored; the operation order below is
// the one its scalar f64 path documents (glm-style cofactor inverse, column-by-column
// products), so results agree to the last bit wherever that order is what glam executes and
// to an ulp of binary64 otherwise -- far below the binary32 rounding applied by as_f32().
#pragma once
#include <cmath>

namespace ptl {

struct DVec3 {
    double x = 0, y = 0, z = 0;
    DVec3() = default;
    DVec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    DVec3 operator+(const DVec3& o) const { return {x + o.x, y + o.y, z + o.z}; }
    DVec3 operator-(const DVec3& o) const { return {x - o.x, y - o.y, z - o.z}; }
    DVec3 operator*(double s) const { return {x * s, y * s, z * s}; }
    double dot(const DVec3& o) const { return x * o.x + y * o.y + z * o.z; }
    double length() const { return std::sqrt(dot(*this)); }
    DVec3 normalize() const { return *this * (1.0 / length()); }
    DVec3 cross(const DVec3& o) const { return {y * o.z - o.y * z, z * o.x - o.z * x, x * o.y - o.x * y}; }
    DVec3 lerp(const DVec3& o, double t) const { return *this + (o - *this) * t; }
};

struct DVec4 {
    double x = 0, y = 0, z = 0, w = 0;
    DVec4() = default;
    DVec4(double x_, double y_, double z_, double w_) : x(x_), y(y_), z(z_), w(w_) {}
    DVec4 operator*(double s) const { return {x * s, y * s, z * s, w * s}; }
    DVec4 operator*(const DVec4& o) const { return {x * o.x, y * o.y, z * o.z, w * o.w}; }
    DVec4 operator+(const DVec4& o) const { return {x + o.x, y + o.y, z + o.z, w + o.w}; }
    DVec4 operator-(const DVec4& o) const { return {x - o.x, y - o.y, z - o.z, w - o.w}; }
    double length() const { return std::sqrt(x * x + y * y + z * z + w * w); }
};

struct DQuat {
    double x = 0, y = 0, z = 0, w = 1;
    static DQuat rotation_x(double a) { return {std::sin(a * 0.5), 0, 0, std::cos(a * 0.5)}; }
    static DQuat rotation_y(double a) { return {0, std::sin(a * 0.5), 0, std::cos(a * 0.5)}; }
    static DQuat rotation_z(double a) { return {0, 0, std::sin(a * 0.5), std::cos(a * 0.5)}; }
    // glam 0.13 Quaternion::from_rotation_axes (Mike Day, "Converting a Rotation Matrix to a Quaternion")
    static DQuat from_rotation_axes(const DVec3& xa, const DVec3& ya, const DVec3& za) {
        double m00 = xa.x, m01 = xa.y, m02 = xa.z, m10 = ya.x, m11 = ya.y, m12 = ya.z, m20 = za.x, m21 = za.y, m22 = za.z;
        if (m22 <= 0.0) {
            double dif10 = m11 - m00, omm22 = 1.0 - m22;
            if (dif10 <= 0.0) {
                double four_xsq = omm22 - dif10, inv4x = 0.5 / std::sqrt(four_xsq);
                return {four_xsq * inv4x, (m01 + m10) * inv4x, (m02 + m20) * inv4x, (m12 - m21) * inv4x};
            }
            double four_ysq = omm22 + dif10, inv4y = 0.5 / std::sqrt(four_ysq);
            return {(m01 + m10) * inv4y, four_ysq * inv4y, (m12 + m21) * inv4y, (m20 - m02) * inv4y};
        }
        double sum10 = m11 + m00, opm22 = 1.0 + m22;
        if (sum10 <= 0.0) {
            double four_zsq = opm22 - sum10, inv4z = 0.5 / std::sqrt(four_zsq);
            return {(m02 + m20) * inv4z, (m12 + m21) * inv4z, four_zsq * inv4z, (m01 - m10) * inv4z};
        }
        double four_wsq = opm22 + sum10, inv4w = 0.5 / std::sqrt(four_wsq);
        return {(m12 - m21) * inv4w, (m20 - m02) * inv4w, (m01 - m10) * inv4w, four_wsq * inv4w};
    }
    double dot(const DQuat& o) const { return x * o.x + y * o.y + z * o.z + w * o.w; }
    // glam 0.13 Quaternion::lerp: shortest-arc linear blend, then normalize (x * (1 / length))
    DQuat lerp(const DQuat& end, double s) const {
        double bias = dot(end) >= 0.0 ? 1.0 : -1.0;
        DQuat q{x + (end.x * bias - x) * s, y + (end.y * bias - y) * s, z + (end.z * bias - z) * s, w + (end.w * bias - w) * s};
        double inv = 1.0 / std::sqrt(q.dot(q));
        return {q.x * inv, q.y * inv, q.z * inv, q.w * inv};
    }
    DQuat operator*(const DQuat& o) const {
        return {w * o.x + x * o.w + y * o.z - z * o.y, w * o.y - x * o.z + y * o.w + z * o.x,
                w * o.z + x * o.y - y * o.x + z * o.w, w * o.w - x * o.x - y * o.y - z * o.z};
    }
};

// Column-major 4x4, c[k] is column k (glam x_axis..w_axis).
struct DMat4 {
    DVec4 c[4];
    static DMat4 identity() {
        DMat4 m;
        m.c[0] = {1, 0, 0, 0};
        m.c[1] = {0, 1, 0, 0};
        m.c[2] = {0, 0, 1, 0};
        m.c[3] = {0, 0, 0, 1};
        return m;
    }
    static DMat4 from_cols(const DVec4& a, const DVec4& b, const DVec4& cc, const DVec4& d) {
        DMat4 m;
        m.c[0] = a; m.c[1] = b; m.c[2] = cc; m.c[3] = d;
        return m;
    }
    static DMat4 from_scale_rotation_translation(const DVec3& s, const DQuat& q, const DVec3& t) {
        double x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
        double xx = q.x * x2, xy = q.x * y2, xz = q.x * z2;
        double yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
        double wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
        DVec4 xa(1.0 - (yy + zz), xy + wz, xz - wy, 0.0);
        DVec4 ya(xy - wz, 1.0 - (xx + zz), yz + wx, 0.0);
        DVec4 za(xz + wy, yz - wx, 1.0 - (xx + yy), 0.0);
        return from_cols(xa * s.x, ya * s.y, za * s.z, DVec4(t.x, t.y, t.z, 1.0));
    }
    double determinant() const {  // glam 0.13 Matrix4x4::determinant
        double m00 = c[0].x, m01 = c[0].y, m02 = c[0].z, m03 = c[0].w, m10 = c[1].x, m11 = c[1].y, m12 = c[1].z, m13 = c[1].w;
        double m20 = c[2].x, m21 = c[2].y, m22 = c[2].z, m23 = c[2].w, m30 = c[3].x, m31 = c[3].y, m32 = c[3].z, m33 = c[3].w;
        double a2323 = m22 * m33 - m23 * m32, a1323 = m21 * m33 - m23 * m31, a1223 = m21 * m32 - m22 * m31;
        double a0323 = m20 * m33 - m23 * m30, a0223 = m20 * m32 - m22 * m30, a0123 = m20 * m31 - m21 * m30;
        return m00 * (m11 * a2323 - m12 * a1323 + m13 * a1223) - m01 * (m10 * a2323 - m12 * a0323 + m13 * a0223) +
               m02 * (m10 * a1323 - m11 * a0323 + m13 * a0123) - m03 * (m10 * a1223 - m11 * a0223 + m12 * a0123);
    }
    // glam 0.13 Mat4::to_scale_rotation_translation: scale = axis lengths (x negated for a mirrored basis), rotation from
    // the axes divided by the scale, translation = w axis
    void to_scale_rotation_translation(DVec3* scale, DQuat* rotation, DVec3* translation) const {
        double det = determinant();
        double sign = std::isnan(det) ? det : (std::signbit(det) ? -1.0 : 1.0);  // f64::signum
        *scale = DVec3(c[0].length() * sign, c[1].length(), c[2].length());
        double ix = 1.0 / scale->x, iy = 1.0 / scale->y, iz = 1.0 / scale->z;
        *rotation = DQuat::from_rotation_axes(DVec3(c[0].x * ix, c[0].y * ix, c[0].z * ix), DVec3(c[1].x * iy, c[1].y * iy, c[1].z * iy),
                                              DVec3(c[2].x * iz, c[2].y * iz, c[2].z * iz));
        *translation = DVec3(c[3].x, c[3].y, c[3].z);
    }
    DVec4 mul_vec4(const DVec4& v) const {
        DVec4 r = c[0] * v.x;
        r = c[1] * v.y + r;
        r = c[2] * v.z + r;
        r = c[3] * v.w + r;
        return r;
    }
    DMat4 operator*(const DMat4& o) const { return from_cols(mul_vec4(o.c[0]), mul_vec4(o.c[1]), mul_vec4(o.c[2]), mul_vec4(o.c[3])); }
    DMat4 operator*(double s) const { return from_cols(c[0] * s, c[1] * s, c[2] * s, c[3] * s); }
    DMat4 inverse() const {
        double m00 = c[0].x, m01 = c[0].y, m02 = c[0].z, m03 = c[0].w;
        double m10 = c[1].x, m11 = c[1].y, m12 = c[1].z, m13 = c[1].w;
        double m20 = c[2].x, m21 = c[2].y, m22 = c[2].z, m23 = c[2].w;
        double m30 = c[3].x, m31 = c[3].y, m32 = c[3].z, m33 = c[3].w;
        double coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
        double coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
        double coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
        double coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
        double coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
        double coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
        DVec4 fac0(coef00, coef00, coef02, coef03), fac1(coef04, coef04, coef06, coef07);
        DVec4 fac2(coef08, coef08, coef10, coef11), fac3(coef12, coef12, coef14, coef15);
        DVec4 fac4(coef16, coef16, coef18, coef19), fac5(coef20, coef20, coef22, coef23);
        DVec4 vec0(m10, m00, m00, m00), vec1(m11, m01, m01, m01), vec2(m12, m02, m02, m02), vec3(m13, m03, m03, m03);
        DVec4 inv0 = vec1 * fac0 - vec2 * fac1 + vec3 * fac2;
        DVec4 inv1 = vec0 * fac0 - vec2 * fac3 + vec3 * fac4;
        DVec4 inv2 = vec0 * fac1 - vec1 * fac3 + vec3 * fac5;
        DVec4 inv3 = vec0 * fac2 - vec1 * fac4 + vec2 * fac5;
        DVec4 sign_a(1.0, -1.0, 1.0, -1.0), sign_b(-1.0, 1.0, -1.0, 1.0);
        DMat4 inv = from_cols(inv0 * sign_a, inv1 * sign_b, inv2 * sign_a, inv3 * sign_b);
        DVec4 col0(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
        DVec4 dot0 = c[0] * col0;
        double dot1 = dot0.x + dot0.y + dot0.z + dot0.w;
        return inv * (1.0 / dot1);
    }
    DMat4 operator+(const DMat4& o) const { return from_cols(c[0] + o.c[0], c[1] + o.c[1], c[2] + o.c[2], c[3] + o.c[3]); }
    bool same_bits(const DMat4& o) const {
        for (int k = 0; k < 4; ++k)
            if (c[k].x != o.c[k].x || c[k].y != o.c[k].y || c[k].z != o.c[k].z || c[k].w != o.c[k].w) return false;
        return true;
    }
    // Principal square root by the Denman-Beavers iteration Y <- (Y + Z^-1)/2, Z <- (Z + Y^-1)/2 from (A, I): Y -> A^(1/2).
    // Stops when Y repeats bit for bit (quadratic convergence: ~8 rounds) or after 64 rounds; *ok = Y*Y reproduces A to 1e-9.
    DMat4 sqrt_principal(bool* ok) const {
        DMat4 y = *this, z = identity();
        for (int round = 0; round < 64; ++round) {
            DMat4 yn = (y + z.inverse()) * 0.5, zn = (z + y.inverse()) * 0.5;
            bool done = yn.same_bits(y);
            y = yn;
            z = zn;
            if (done) break;
        }
        DMat4 sq = y * y;
        double err = 0.0;
        for (int k = 0; k < 4; ++k) {
            DVec4 d = sq.c[k] - c[k];
            err += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        }
        *ok = err < 1e-9;  // also false for NaN (a matrix without a real square root diverges)
        return y;
    }
    // as_f32(): one round-to-nearest per element, column-major
    void to_f32(float out[16]) const {
        for (int k = 0; k < 4; ++k) {
            out[4 * k + 0] = (float)c[k].x;
            out[4 * k + 1] = (float)c[k].y;
            out[4 * k + 2] = (float)c[k].z;
            out[4 * k + 3] = (float)c[k].w;
        }
    }
};

}  // namespace ptl
