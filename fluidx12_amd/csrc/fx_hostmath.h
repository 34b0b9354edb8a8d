// fx_hostmath.h -- the few DirectXMath operations Fluid::UpdateFrame needs, in scalar fp32.
// Row-vector convention (v' = v * M), row-major storage, left-handed projection -- the
// conventions of the reference's host code (FluidX12/Content/Fluid.cpp:283-346).
#pragma once
#include <cmath>
#include <cstring>

namespace fx {

struct Vec3 { float x, y, z; };

struct Mat4 {
	float m[4][4];

	static Mat4 identity()
	{
		Mat4 r;
		std::memset(r.m, 0, sizeof r.m);
		r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f;
		return r;
	}
	static Mat4 scaling(float sx, float sy, float sz)
	{
		Mat4 r = identity();
		r.m[0][0] = sx; r.m[1][1] = sy; r.m[2][2] = sz;
		return r;
	}
	static Mat4 from(const float* p) { Mat4 r; std::memcpy(r.m, p, sizeof r.m); return r; }

	Mat4 operator*(const Mat4& o) const
	{
		Mat4 r;
		for (int i = 0; i < 4; ++i)
			for (int j = 0; j < 4; ++j) {
				float acc = m[i][0] * o.m[0][j];
				acc = std::fmaf(m[i][1], o.m[1][j], acc);
				acc = std::fmaf(m[i][2], o.m[2][j], acc);
				acc = std::fmaf(m[i][3], o.m[3][j], acc);
				r.m[i][j] = acc;
			}
		return r;
	}

	// 3x3 minor determinant helpers for the adjugate
	float minor3(int r0, int r1, int r2, int c0, int c1, int c2) const
	{
		return m[r0][c0] * (m[r1][c1] * m[r2][c2] - m[r1][c2] * m[r2][c1])
			- m[r0][c1] * (m[r1][c0] * m[r2][c2] - m[r1][c2] * m[r2][c0])
			+ m[r0][c2] * (m[r1][c0] * m[r2][c1] - m[r1][c1] * m[r2][c0]);
	}

	// adjugate / determinant (XMMatrixInverse multiplies the adjugate by 1/det)
	Mat4 inverse() const
	{
		Mat4 adj;
		static const int others[4][3] = { {1,2,3},{0,2,3},{0,1,3},{0,1,2} };
		for (int i = 0; i < 4; ++i)
			for (int j = 0; j < 4; ++j) {
				const int* r = others[j];   // cofactor of element (j, i) goes to adj(i, j)
				const int* c = others[i];
				const float mn = minor3(r[0], r[1], r[2], c[0], c[1], c[2]);
				adj.m[i][j] = ((i + j) & 1) ? -mn : mn;
			}
		float det = 0.0f;
		for (int k = 0; k < 4; ++k) det += m[0][k] * adj.m[k][0];
		const float rdet = 1.0f / det;
		for (int i = 0; i < 4; ++i)
			for (int j = 0; j < 4; ++j) adj.m[i][j] *= rdet;
		return adj;
	}

	// XMStoreFloat3x4: the first three columns, each as a row of four
	void store3x4(float out[12]) const
	{
		for (int c = 0; c < 3; ++c)
			for (int r = 0; r < 4; ++r) out[c * 4 + r] = m[r][c];
	}

	// XMVector3TransformCoord: (v,1) * M, divided by w
	Vec3 transform_coord(const Vec3& v) const
	{
		float h[4];
		for (int j = 0; j < 4; ++j)
			h[j] = std::fmaf(v.z, m[2][j], std::fmaf(v.y, m[1][j], v.x * m[0][j])) + m[3][j];
		return Vec3{ h[0] / h[3], h[1] / h[3], h[2] / h[3] };
	}
	// XMVector3Transform: (v,1) * M without the divide; returns component `j`
	float transform_comp(const float v[3], int j) const
	{
		return std::fmaf(v[2], m[2][j], std::fmaf(v[1], m[1][j], v[0] * m[0][j])) + m[3][j];
	}
};

}  // namespace fx
