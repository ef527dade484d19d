// dbvh.h — BVH2 traversal with a per-lane short stack in LDS (wave64; stack[d][lane] layout => conflict-free banks).
// Replaces ocarina::Accel trace_closest / trace_occlusion (base/mgr/geometry.cpp:168-185; OptiX in the reference).
// Hit selection rule (shared with the oracle): valid hits 0 < t < t_max, smallest t wins, equal t resolved towards
// the smaller (inst, prim).  Triangle test: Moeller-Trumbore on world-space vertices, IEEE float32, no contraction.
#pragma once
#include "dpath.h"

namespace vmkd {

// Traversal stack: LDS only, stack[level][thread] (stride = block size, so consecutive lanes hit consecutive banks).
// 40 levels x 4 B = 10 KB per wave -> 16 waves per CU fit in 160 KB.  The depth of a built tree is checked against
// kStackDepth at build time.  (Measured alternatives on classroom: spilling deep levels to a per-lane private array
// costs 28 % — 1920 -> 1373 Mrays/s — and 48 LDS levels at 2 blocks/CU cost 41 %.)
constexpr int kStackDepth = 40;

struct Hit { uint32_t inst, prim, tri; V2 bary; };

VD bool intersect_tri(const vmk_tri_pos *tp, V3 o, V3 d, float *t_out, float *u_out, float *v_out, uint32_t *inst_out, uint32_t *prim_out) {
    // 48 B record as three 16 B loads
    const float4 *q = reinterpret_cast<const float4 *>(tp);
    float4 a = q[0], b = q[1], c = q[2];
    V3 p0 = {a.x, a.y, a.z}, p1 = {a.w, b.x, b.y}, p2 = {b.z, b.w, c.x};
    *inst_out = f2u(c.y); *prim_out = f2u(c.z);
    V3 e1 = p1 - p0, e2 = p2 - p0;
    V3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    if (det == 0.f) return false;
    float inv = 1.f / det;
    V3 tvec = o - p0;
    float u = dot(tvec, pvec) * inv;
    if (!(u >= 0.f && u <= 1.f)) return false;
    V3 qvec = cross(tvec, e1);
    float v = dot(d, qvec) * inv;
    if (!(v >= 0.f && u + v <= 1.f)) return false;
    *t_out = dot(e2, qvec) * inv;
    *u_out = u; *v_out = v;
    return true;
}

// conservative slab test; NaN slabs (0 * inf) are ignored by the min/max (IEEE minNum/maxNum on v_min/v_max_f32)
VD bool hit_box(const float *bmin, const float *bmax, V3 o, V3 inv, float t_far, float *t_near_out) {
    float tx0 = (bmin[0] - o.x) * inv.x, tx1 = (bmax[0] - o.x) * inv.x;
    float ty0 = (bmin[1] - o.y) * inv.y, ty1 = (bmax[1] - o.y) * inv.y;
    float tz0 = (bmin[2] - o.z) * inv.z, tz1 = (bmax[2] - o.z) * inv.z;
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx0, tx1), __builtin_fminf(ty0, ty1)), __builtin_fmaxf(__builtin_fminf(tz0, tz1), 0.f));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx0, tx1), __builtin_fmaxf(ty0, ty1)), __builtin_fminf(__builtin_fmaxf(tz0, tz1), t_far));
    *t_near_out = tn;
    return tn * 0.999999f <= tf * 1.000001f;
}

// `stack` points at this lane's column of the block's LDS stack; consecutive levels are `stride` words apart.
template<bool ANY_HIT>
VD bool traverse(const DScene &S, const Ray &r, uint32_t *stack, int stride, Hit &hit, DCounters &cnt) {
    V3 inv = {1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z};
    float best_t = r.t_max;
    hit.inst = VMK_INVALID; hit.prim = VMK_INVALID; hit.tri = VMK_INVALID; hit.bary = {0.f, 0.f};
    if (S.n_tris == 0) return false;
    constexpr int32_t kDone = 0x7fffffff;
    int sp = 0;
    int32_t cur = S.root;
    uint32_t nn = 0, nt = 0;
    bool found = false;
    // "while-while": all lanes of the wave first descend through internal nodes, then the lanes that reached a leaf test
    // their triangles together — fewer serialised node/leaf branches per wave (+14 % Mrays/s on classroom).
    while (cur != kDone) {
        while (cur >= 0 && cur != kDone) {
            const float4 *q = reinterpret_cast<const float4 *>(S.nodes + cur);
            float4 n0 = q[0], n1 = q[1], n2 = q[2], n3 = q[3];
            ++nn;
            float lmin[3] = {n0.x, n0.y, n0.z}, lmax[3] = {n0.w, n1.x, n1.y};
            float rmin[3] = {n1.z, n1.w, n2.x}, rmax[3] = {n2.y, n2.z, n2.w};
            int32_t left = (int32_t) f2u(n3.x), right = (int32_t) f2u(n3.y);
            float tl, tr;
            bool hl = hit_box(lmin, lmax, r.o, inv, best_t, &tl);
            bool hr = hit_box(rmin, rmax, r.o, inv, best_t, &tr);
            if (hl && hr) {
                bool left_first = tl <= tr;
                if (sp < kStackDepth) { stack[sp * stride] = (uint32_t) (left_first ? right : left); ++sp; }
                cur = left_first ? left : right;
            } else if (hl) cur = left;
            else if (hr) cur = right;
            else if (sp > 0) { --sp; cur = (int32_t) stack[sp * stride]; }
            else cur = kDone;
        }
        if (cur != kDone) {
            uint32_t v = ~(uint32_t) cur;
            uint32_t first = v & kLeafFirstMask, count = (v >> 28) + 1u;
            for (uint32_t i = 0; i < count; ++i) {
                const vmk_tri_pos *tp = S.tri_pos + first + i;
                float t, u, w;
                uint32_t inst, prim;
                ++nt;
                if (!intersect_tri(tp, r.o, r.d, &t, &u, &w, &inst, &prim)) continue;
                if (!(t > 0.f && t < r.t_max)) continue;
                if constexpr (ANY_HIT) { found = true; break; }
                bool better = !found ? (t <= best_t) : (t < best_t || (t == best_t && (inst < hit.inst || (inst == hit.inst && prim < hit.prim))));
                if (better) { best_t = t; hit.inst = inst; hit.prim = prim; hit.tri = first + i; hit.bary = {u, w}; found = true; }
            }
            if (ANY_HIT && found) break;
            if (sp > 0) { --sp; cur = (int32_t) stack[sp * stride]; }
            else cur = kDone;
        }
    }
    cnt.nodes += nn; cnt.tris += nt;
    return found;
}

}// namespace vmkd
