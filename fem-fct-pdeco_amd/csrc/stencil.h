// Structured right-diagonal P1 mesh of a square: stencil/triangle tables shared by
// the host pattern builder and the device assembly kernels.
//
// Mesh: dolfin RectangleMesh(Point(a1,a1),Point(a2,a2),nc,nc), diagonal "right":
// cell (cx,cy) -> triangles L=(v0,v1,v3), U=(v0,v2,v3) with v0=(cx,cy), v1=(cx+1,cy),
// v2=(cx,cy+1), v3=(cx+1,cy+1)   (pinned by the shipped
// FEniCS trajectory through the oracle).
//
// ELL slots of row P=(ix,iy):  0 self | 1 E(+1,0) | 2 NE(+1,+1) | 3 N(0,+1)
//                              4 W(-1,0) | 5 SW(-1,-1) | 6 S(0,-1)
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define FEMFCT_HD __host__ __device__ __forceinline__
#else
#define FEMFCT_HD inline
#endif

#define STENCIL_W 7

FEMFCT_HD int stencil_dx(int s) { const int t[7] = {0, 1, 1, 0, -1, -1, 0}; return t[s]; }
FEMFCT_HD int stencil_dy(int s) { const int t[7] = {0, 0, 1, 1, 0, -1, -1}; return t[s]; }
FEMFCT_HD int stencil_opp(int s) { const int t[7] = {0, 4, 5, 6, 1, 2, 3}; return t[s]; }

// The six triangles around P, counter-clockwise starting in the cell (ix,iy).
//   type 0 = L (v0,v1,v3): grad(lambda)*h = (-1,0), (1,-1), (0,1)
//   type 1 = U (v0,v2,v3): grad(lambda)*h = (0,-1), (-1,1), (1,0)
struct TriInfo {
    int type;      // 0: L, 1: U
    int pl;        // local index of P in the triangle
    int cdx, cdy;  // cell = (ix+cdx, iy+cdy)
    int slot[3];   // ELL slot (in row P) of the triangle's local nodes
};

FEMFCT_HD TriInfo tri_info(int t) {
    const TriInfo T[6] = {
        {0, 0, 0, 0, {0, 1, 2}},     // (P, E, NE)
        {1, 0, 0, 0, {0, 3, 2}},     // (P, N, NE)
        {0, 1, -1, 0, {4, 0, 3}},    // (W, P, N)
        {1, 1, 0, -1, {6, 0, 1}},    // (S, P, E)
        {0, 2, -1, -1, {5, 6, 0}},   // (SW, S, P)
        {1, 2, -1, -1, {5, 4, 0}},   // (SW, W, P)
    };
    return T[t];
}

// h * grad(lambda_k) for local node k of a triangle of the given type
FEMFCT_HD double tri_gx(int type, int k) { const double g[2][3] = {{-1, 1, 0}, {0, -1, 1}}; return g[type][k]; }
FEMFCT_HD double tri_gy(int type, int k) { const double g[2][3] = {{0, -1, 1}, {-1, 1, 0}}; return g[type][k]; }

// offset (in cells) of local node k of a triangle of `type` from the cell origin v0
FEMFCT_HD int tri_nx(int type, int k) { const int o[2][3] = {{0, 1, 1}, {0, 0, 1}}; return o[type][k]; }
FEMFCT_HD int tri_ny(int type, int k) { const int o[2][3] = {{0, 0, 1}, {0, 1, 1}}; return o[type][k]; }

// 6-point degree-4 rule (FIAT "default" scheme for degree 4; barycentric, weights sum to 1).
// Exact for every polynomial form of the reference (degree <= 4); it is also the rule
// FEniCS uses for the forward chemotaxis exp-form (helpers.py:1350-1351).
#define QUAD6_A 0.091576213509771
#define QUAD6_B 0.445948490915965
#define QUAD6_WA 0.109951743655322
#define QUAD6_WB 0.223381589678011
FEMFCT_HD double quad6_l(int q, int k) {
    // point q: barycentric coordinate k
    const double a = QUAD6_A, b = QUAD6_B;
    const double L[6][3] = {{1 - 2 * a, a, a}, {a, 1 - 2 * a, a}, {a, a, 1 - 2 * a},
                            {1 - 2 * b, b, b}, {b, 1 - 2 * b, b}, {b, b, 1 - 2 * b}};
    return L[q][k];
}
FEMFCT_HD double quad6_w(int q) { return q < 3 ? QUAD6_WA : QUAD6_WB; }

// 7-point degree-5 Radon rule (FIAT "default" scheme for degree 5)
FEMFCT_HD double quad7_l(int q, int k) {
    const double c = 0.10128650732345633, d = 0.47014206410511505, t = 1.0 / 3.0;
    const double L[7][3] = {{t, t, t}, {1 - 2 * c, c, c}, {c, 1 - 2 * c, c}, {c, c, 1 - 2 * c},
                            {1 - 2 * d, d, d}, {d, 1 - 2 * d, d}, {d, d, 1 - 2 * d}};
    return L[q][k];
}
FEMFCT_HD double quad7_w(int q) {
    return q == 0 ? 0.225 : (q < 4 ? 0.12593918054482717 : 0.13239415278850616);
}
