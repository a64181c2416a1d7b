// box_geom.h -- rotated-box BEV geometry shared by iou3d_nms.hip (pairwise matrices, NMS) and atss.hip (ATSS assigner).
// Restates cpd/ops/iou3d_nms/src/iou3d_nms_kernel.cu:35-234 with the reference's fp32 operation order (edge x edge crossings
// i=0..3 x j=0..3, corners interleaved b-in-a / a-in-b, centroid, bubble sort by atan2, shoelace fan), so degenerate cases
// (MARGIN corners, parallel edges) agree. __host__ __device__: boxes_iou_bev_cpu (iou3d_cpu.cpp:232-252) runs the same code.
#pragma once
#include <math.h>

#include "common.h"

#ifndef IOU_EPS
#define IOU_EPS 1e-8f
#endif

namespace {

struct P2 { float x, y; };

__host__ __device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

__host__ __device__ __forceinline__ bool rect_cross(P2 p1, P2 p2, P2 q1, P2 q2) {
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

struct BoxG {      // per-box derived geometry, computed once per box per pair
    float b[7];
    float cs, sn;  // cos/sin(heading)
    float ncs, nsn;  // cos/sin(-heading) as the reference evaluates them for the corner test
    P2 c[5];
};

__host__ __device__ __forceinline__ void box_setup(const float *box, BoxG &g) {
#pragma unroll
    for (int k = 0; k < 7; ++k) g.b[k] = box[k];
    const float hx = box[3] / 2, hy = box[4] / 2;
    const float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
    g.cs = cosf(box[6]);
    g.sn = sinf(box[6]);
    g.ncs = cosf(-box[6]);
    g.nsn = sinf(-box[6]);
    const float rx[4] = {x1, x2, x2, x1}, ry[4] = {y1, y1, y2, y2};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float dx = rx[k] - box[0], dy = ry[k] - box[1];
        g.c[k].x = dx * g.cs + dy * (-g.sn) + box[0];
        g.c[k].y = dx * g.sn + dy * g.cs + box[1];
    }
    g.c[4] = g.c[0];
}

__host__ __device__ __forceinline__ bool in_box2d(const BoxG &g, P2 p) {
    const float margin = 1e-2f;
    const float rx = (p.x - g.b[0]) * g.ncs + (p.y - g.b[1]) * (-g.nsn);
    const float ry = (p.x - g.b[0]) * g.nsn + (p.y - g.b[1]) * g.ncs;
    return fabsf(rx) < g.b[3] / 2 + margin && fabsf(ry) < g.b[4] / 2 + margin;
}

__host__ __device__ __forceinline__ bool seg_intersection(P2 p1, P2 p0, P2 q1, P2 q0, P2 &ans) {
    if (!rect_cross(p0, p1, q0, q1)) return false;
    const float s1 = cross3(q0, p1, p0);
    const float s2 = cross3(p1, q1, p0);
    const float s3 = cross3(p0, q1, q0);
    const float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > IOU_EPS) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__host__ __device__ inline float box_overlap_g(const BoxG &A, const BoxG &B) {
    P2 pts[16];
    float ang[16];
    float cx = 0.f, cy = 0.f;
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 x;
            if (seg_intersection(A.c[i + 1], A.c[i], B.c[j + 1], B.c[j], x)) {
                cx = cx + x.x;
                cy = cy + x.y;
                pts[cnt++] = x;
            }
        }
    for (int k = 0; k < 4; ++k) {
        if (in_box2d(A, B.c[k])) {
            cx = cx + B.c[k].x; cy = cy + B.c[k].y;
            pts[cnt++] = B.c[k];
        }
        if (in_box2d(B, A.c[k])) {
            cx = cx + A.c[k].x; cy = cy + A.c[k].y;
            pts[cnt++] = A.c[k];
        }
    }
    if (cnt == 0) return 0.f;  // reference: 0/0 centroid, empty loops, |0|/2
    cx /= cnt;
    cy /= cnt;
    // the reference re-evaluates atan2 inside every comparison; the angles are pure functions of
    // the points, so evaluating them once and carrying them through the swaps is identical.
    for (int k = 0; k < cnt; ++k) ang[k] = atan2f(pts[k].y - cy, pts[k].x - cx);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                P2 tp = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = tp;
                float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const float ux = pts[k].x - pts[0].x, uy = pts[k].y - pts[0].y;
        const float vx = pts[k + 1].x - pts[0].x, vy = pts[k + 1].y - pts[0].y;
        area += ux * vy - uy * vx;
    }
    return fabsf(area) / 2.0f;
}

__host__ __device__ __forceinline__ float iou_bev_g(const BoxG &A, const BoxG &B) {
    const float sa = A.b[3] * A.b[4], sb = B.b[3] * B.b[4];
    const float so = box_overlap_g(A, B);
    return so / fmaxf(sa + sb - so, IOU_EPS);
}

__host__ __device__ __forceinline__ float iou_normal_g(const float *a, const float *b) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float inter = width * height;
    return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, IOU_EPS);
}

}  // namespace
