// mc_cell.h -- one Lewiner MC33 cell: which tiling row applies, how many triangles, does it use
// the cell-centre vertex.  Pure function of the 8 corner values; compiled for the device (mc.hip)
// and, for the CPU-side unit tests, for the host (host_shim.cpp).
//
// Replaces, per cell, what skimage.measure.marching_cubes does inside the reference call
// surface_extractor.py:55 (Lewiner variant, skimage/measure/_marching_cubes_lewiner.py:280-349).
// The data-parallel design differs from the serial original: no face layers and no first-touch
// numbering -- a vertex is identified by the edge it sits on (owner voxel + slot), and the cell
// only has to report its triangle list.
#pragma once

#ifndef MC_FN
#define MC_FN static inline
#endif

#include "mc_luts.h"

#define MC_EPS 0x1p-52  // np.spacing(1.0): the epsilon of the compiled Lewiner core

// corner quadruples (A,B,C,D) of cube faces 1..6 (index 0 unused)
MC_LUT_QUAL signed char MC_FACE_Q[7][4] = {{0, 0, 0, 0}, {0, 4, 5, 1}, {1, 5, 6, 2}, {2, 6, 7, 3},
                                           {3, 7, 4, 0}, {0, 3, 2, 1}, {4, 7, 6, 5}};
// per reference edge: its corners (a,b), then the three edges parallel to it as (from,to) pairs
MC_LUT_QUAL signed char MC_PAR_EDGES[12][8] = {
    {0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
    {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2}, {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0},
    {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};

// v[i] with a run-time i, as a select chain so that v[] stays in registers on the device
#define MC_V8P double v0, double v1, double v2, double v3, double v4, double v5, double v6, double v7
#define MC_V8A v0, v1, v2, v3, v4, v5, v6, v7
// The corner values travel as eight scalars, not as a pointer: on a pointer the optimizer folds the select chain below
// into one load at a run-time address, and that single indexed load pins the caller's whole cell record in scratch
// memory (64 B of stores per lane in every kernel that evaluates a cell).
MC_FN double mc_pick(MC_V8P, int i)
{
    double r = v0;
    r = i == 1 ? v1 : r; r = i == 2 ? v2 : r; r = i == 3 ? v3 : r; r = i == 4 ? v4 : r;
    r = i == 5 ? v5 : r; r = i == 6 ? v6 : r; r = i == 7 ? v7 : r;
    return r;
}

struct McTiling {
    const signed char *tris;  // 3*ntri edge ids (0..11 cube edges, 12 = centre vertex)
    int ntri;
    int centre;               // 1 if any entry is 12
};

// "does the surface cross this face with the positive corners joined": sign of A*C - B*D.
MC_FN int mc_test_face(MC_V8P, int face)
{
    int a = face < 0 ? -face : face;
    double A = mc_pick(MC_V8A, MC_FACE_Q[a][0]), B = mc_pick(MC_V8A, MC_FACE_Q[a][1]), C = mc_pick(MC_V8A, MC_FACE_Q[a][2]),
           D = mc_pick(MC_V8A, MC_FACE_Q[a][3]);
    double d = A * C - B * D;
    if (d > -MC_EPS && d < MC_EPS) return face >= 0;
    return (double)face * A * d >= 0;
}

// interior ("tunnel") test.  kase in {4,6,7,10,12,13}; s = LUT sign; refedge only for 6/7/12/13.
MC_FN int mc_test_interior(MC_V8P, int kase, int refedge, int s)
{
    double t, At = 0.0, Bt, Ct, Dt;
    if (kase == 4 || kase == 10) {
        double a = (v4 - v0) * (v6 - v2) - (v7 - v3) * (v5 - v1);
        double b = v2 * (v4 - v0) + v0 * (v6 - v2) - v1 * (v7 - v3) - v3 * (v5 - v1);
        t = -b / (2 * a + MC_EPS);
        if (t < 0 || t > 1) return s > 0;
        At = v0 + (v4 - v0) * t;
        Bt = v3 + (v7 - v3) * t;
        Ct = v2 + (v6 - v2) * t;
        Dt = v1 + (v5 - v1) * t;
    } else {
        if (refedge < 0 || refedge > 11) return s < 0;
        const signed char *q = MC_PAR_EDGES[refedge];
        double va = mc_pick(MC_V8A, q[0]), vb = mc_pick(MC_V8A, q[1]);
        double b0 = mc_pick(MC_V8A, q[2]), b1 = mc_pick(MC_V8A, q[3]), c0 = mc_pick(MC_V8A, q[4]), c1 = mc_pick(MC_V8A, q[5]);
        double d0 = mc_pick(MC_V8A, q[6]), d1 = mc_pick(MC_V8A, q[7]);
        t = va / (va - vb + MC_EPS);
        Bt = b0 + (b1 - b0) * t;
        Ct = c0 + (c1 - c0) * t;
        Dt = d0 + (d1 - d0) * t;
    }
    int test = (At >= 0.0 ? 1 : 0) + (Bt >= 0.0 ? 2 : 0) + (Ct >= 0.0 ? 4 : 0) + (Dt >= 0.0 ? 8 : 0);
    switch (test) {
    case 0: case 1: case 2: case 3: case 4: case 6: case 8: case 9: case 12:
        return s > 0;
    // 5 / 10: a failed inner condition yields 0 in the compiled core irrespective of s
    // (pinned by tests/golden/mc_cells.npz, case 4 configs 4-7).
    case 5: return (At * Ct - Bt * Dt < MC_EPS) ? (s > 0) : 0;
    case 10: return (At * Ct - Bt * Dt >= MC_EPS) ? (s > 0) : 0;
    default: return s < 0;
    }
}

#define MC_ROW(T, c) (LUT_##T + (c) * LUT_##T##_D1)
#define MC_ROW2(T, c, s) (LUT_##T + ((c) * LUT_##T##_D1 + (s)) * LUT_##T##_D2)
#define MC_SET(P, N, C) do { r.tris = (P); r.ntri = (N); r.centre = (C); } while (0)

// v[i] = corner value minus iso level, Lewiner corner order; index = sum 2^i [v[i] > 0].
MC_FN McTiling mc_cell_tiling(const double *v, int index)
{
    const double v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], v4 = v[4], v5 = v[5], v6 = v[6], v7 = v[7];
    McTiling r;
    r.tris = LUT_TILING1; r.ntri = 0; r.centre = 0;
    int kase = LUT_CASES[2 * index], c = LUT_CASES[2 * index + 1];
    switch (kase) {
    case 1: MC_SET(MC_ROW(TILING1, c), 1, 0); break;
    case 2: MC_SET(MC_ROW(TILING2, c), 2, 0); break;
    case 5: MC_SET(MC_ROW(TILING5, c), 3, 0); break;
    case 8: MC_SET(MC_ROW(TILING8, c), 2, 0); break;
    case 9: MC_SET(MC_ROW(TILING9, c), 4, 0); break;
    case 11: MC_SET(MC_ROW(TILING11, c), 4, 0); break;
    case 14: MC_SET(MC_ROW(TILING14, c), 4, 0); break;
    case 3:
        if (mc_test_face(MC_V8A, LUT_TEST3[c])) MC_SET(MC_ROW(TILING3_2, c), 4, 0);
        else MC_SET(MC_ROW(TILING3_1, c), 2, 0);
        break;
    case 4:
        if (mc_test_interior(MC_V8A, 4, -1, LUT_TEST4[c])) MC_SET(MC_ROW(TILING4_1, c), 2, 0);
        else MC_SET(MC_ROW(TILING4_2, c), 6, 0);
        break;
    case 6:
        if (mc_test_face(MC_V8A, LUT_TEST6[c * 3])) MC_SET(MC_ROW(TILING6_2, c), 5, 0);
        else if (mc_test_interior(MC_V8A, 6, LUT_TEST6[c * 3 + 2], LUT_TEST6[c * 3 + 1])) MC_SET(MC_ROW(TILING6_1_1, c), 3, 0);
        else MC_SET(MC_ROW(TILING6_1_2, c), 9, 1);
        break;
    case 7: {
        int sub = (mc_test_face(MC_V8A, LUT_TEST7[c * 5]) ? 1 : 0) + (mc_test_face(MC_V8A, LUT_TEST7[c * 5 + 1]) ? 2 : 0) +
                  (mc_test_face(MC_V8A, LUT_TEST7[c * 5 + 2]) ? 4 : 0);
        switch (sub) {
        case 0: MC_SET(MC_ROW(TILING7_1, c), 3, 0); break;
        case 1: MC_SET(MC_ROW2(TILING7_2, c, 0), 5, 0); break;
        case 2: MC_SET(MC_ROW2(TILING7_2, c, 1), 5, 0); break;
        case 3: MC_SET(MC_ROW2(TILING7_3, c, 0), 9, 1); break;
        case 4: MC_SET(MC_ROW2(TILING7_2, c, 2), 5, 0); break;
        case 5: MC_SET(MC_ROW2(TILING7_3, c, 1), 9, 1); break;
        case 6: MC_SET(MC_ROW2(TILING7_3, c, 2), 9, 1); break;
        default:
            if (mc_test_interior(MC_V8A, 7, LUT_TEST7[c * 5 + 4], LUT_TEST7[c * 5 + 3])) MC_SET(MC_ROW(TILING7_4_2, c), 9, 0);
            else MC_SET(MC_ROW(TILING7_4_1, c), 5, 0);
            break;
        }
        break; }
    case 10:
        if (mc_test_face(MC_V8A, LUT_TEST10[c * 3])) {
            if (mc_test_face(MC_V8A, LUT_TEST10[c * 3 + 1])) MC_SET(MC_ROW(TILING10_1_1_, c), 4, 0);
            else MC_SET(MC_ROW(TILING10_2, c), 8, 1);
        } else {
            if (mc_test_face(MC_V8A, LUT_TEST10[c * 3 + 1])) MC_SET(MC_ROW(TILING10_2_, c), 8, 1);
            else if (mc_test_interior(MC_V8A, 10, -1, LUT_TEST10[c * 3 + 2])) MC_SET(MC_ROW(TILING10_1_1, c), 4, 0);
            else MC_SET(MC_ROW(TILING10_1_2, c), 8, 0);
        }
        break;
    case 12:
        if (mc_test_face(MC_V8A, LUT_TEST12[c * 4])) {
            if (mc_test_face(MC_V8A, LUT_TEST12[c * 4 + 1])) MC_SET(MC_ROW(TILING12_1_1_, c), 4, 0);
            else MC_SET(MC_ROW(TILING12_2, c), 8, 1);
        } else {
            if (mc_test_face(MC_V8A, LUT_TEST12[c * 4 + 1])) MC_SET(MC_ROW(TILING12_2_, c), 8, 1);
            else if (mc_test_interior(MC_V8A, 12, LUT_TEST12[c * 4 + 3], LUT_TEST12[c * 4 + 2])) MC_SET(MC_ROW(TILING12_1_1, c), 4, 0);
            else MC_SET(MC_ROW(TILING12_1_2, c), 8, 0);
        }
        break;
    case 13: {
        int sub = 0;
        for (int k = 0; k < 6; k++) sub |= (mc_test_face(MC_V8A, LUT_TEST13[c * 7 + k]) ? 1 : 0) << k;
        sub = LUT_SUBCONFIG13[sub];
        if (sub == 0) MC_SET(MC_ROW(TILING13_1, c), 4, 0);
        else if (sub >= 1 && sub <= 6) MC_SET(MC_ROW2(TILING13_2, c, sub - 1), 6, 0);
        else if (sub >= 7 && sub <= 18) MC_SET(MC_ROW2(TILING13_3, c, sub - 7), 10, 1);
        else if (sub >= 19 && sub <= 22) MC_SET(MC_ROW2(TILING13_4, c, sub - 19), 12, 1);
        else if (sub >= 23 && sub <= 26) {
            int s2 = sub - 23;
            int refedge = LUT_TILING13_5_1[(c * 4 + s2) * 18];
            if (mc_test_interior(MC_V8A, 13, refedge, LUT_TEST13[c * 7 + 6])) MC_SET(MC_ROW2(TILING13_5_1, c, s2), 6, 0);
            else MC_SET(MC_ROW2(TILING13_5_2, c, s2), 10, 0);
        } else if (sub >= 27 && sub <= 38) MC_SET(MC_ROW2(TILING13_3_, c, sub - 27), 10, 1);
        else if (sub >= 39 && sub <= 44) MC_SET(MC_ROW2(TILING13_2_, c, sub - 39), 6, 0);
        else if (sub == 45) MC_SET(MC_ROW(TILING13_1_, c), 4, 0);
        break; }
    default: break;
    }
    return r;
}

// Position of the vertex on a cube edge between corner values va (at offset 0) and vb (at
// offset 1) along the edge axis: weights 1/(eps+|v|), centre-of-mass; returns the offset in [0,1].
MC_FN double mc_edge_offset(double va, double vb)
{
    double wa = 1.0 / (MC_EPS + (va >= 0 ? va : -va));
    double wb = 1.0 / (MC_EPS + (vb >= 0 ? vb : -vb));
    return wb / (wa + wb);
}

// Cell-centre vertex: same weights over the 8 corners, summed in corner order 0..7.
MC_FN void mc_centre_offset(const double *v, double *ox, double *oy, double *oz)
{
    double fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        double w = 1.0 / (MC_EPS + (v[i] >= 0 ? v[i] : -v[i]));
        // corner i sits at (x,y,z) = ((i ^ (i >> 1)) & 1, (i >> 1) & 1, i >> 2); adding 0*w is exact
        if ((i ^ (i >> 1)) & 1) fx += w;
        if ((i >> 1) & 1) fy += w;
        if (i >> 2) fz += w;
        ff += w;
    }
    *ox = fx / ff; *oy = fy / ff; *oz = fz / ff;
}
