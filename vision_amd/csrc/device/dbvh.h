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
#ifndef VMK_STACK_DEPTH
#define VMK_STACK_DEPTH 40
#endif
constexpr int kStackDepth = VMK_STACK_DEPTH;

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

// conservative slab test; NaN slabs (0 * inf) are ignored by the min/max (IEEE minNum/maxNum on v_min/v_max_f32).
// Measured and rejected: t = fma(b, inv, -o*inv) saves 12 VALU ops per node but is not conservative for rays that
// start on a box face (every bounce ray does): its absolute error |o*inv|*2^-24 is unbounded relative to (b-o)*inv.
// The node stores each slab as a (min, max) pair so that both planes of a slab go through one v_pk_add_f32 +
// one v_pk_mul_f32 (gfx950 packed fp32): 6 + 6 instead of 12 + 12 VALU ops per node for the plane distances.
VD bool hit_box(f2v bx, f2v by, f2v bz, V3 o, V3 inv, float t_far, float *t_near_out) {
    f2v tx = (bx - o.x) * inv.x, ty = (by - o.y) * inv.y, tz = (bz - o.z) * inv.z;
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)), __builtin_fmaxf(__builtin_fminf(tz.x, tz.y), 0.f));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)), __builtin_fminf(__builtin_fmaxf(tz.x, tz.y), t_far));
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
            int32_t left = (int32_t) f2u(n3.x), right = (int32_t) f2u(n3.y);
            float tl, tr;
            bool hl = hit_box(f2v{n0.x, n0.y}, f2v{n0.z, n0.w}, f2v{n1.x, n1.y}, r.o, inv, best_t, &tl);
            bool hr = hit_box(f2v{n1.z, n1.w}, f2v{n2.x, n2.y}, f2v{n2.z, n2.w}, r.o, inv, best_t, &tr);
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

// ---------------------------------------------------------------------------------------------------------
// Resumable traversal.  Incoherent rays of one wave need very different numbers of steps (measured on classroom:
// 13-14 of 64 lanes active per VALU instruction in the run-to-completion loop).  The state of a lane's traversal
// therefore lives in a struct that survives leaving the loop: trav_run() returns as soon as fewer than `exit_below`
// lanes of the wave are still traversing, the caller refills the idle lanes (new rays / next path vertex) and calls
// trav_run() again; unfinished lanes simply continue.  Closest-hit and any-hit rays share the one loop.
// ---------------------------------------------------------------------------------------------------------
constexpr int32_t kTravDone = 0x7fffffff;
struct Trav {
    V3 o, d, inv;
    float t_max, best_t;
    int32_t cur;
    int sp;
    bool any_hit, found;
    Hit hit;
    VD bool active() const { return cur != kTravDone; }
};
VD void trav_begin(Trav &T, const DScene &S, const Ray &r, bool any_hit) {
    T.o = r.o; T.d = r.d; T.inv = {1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z};
    T.t_max = r.t_max; T.best_t = r.t_max;
    T.cur = S.n_tris ? S.root : kTravDone; T.sp = 0; T.any_hit = any_hit; T.found = false;
    T.hit.inst = VMK_INVALID; T.hit.prim = VMK_INVALID; T.hit.tri = VMK_INVALID; T.hit.bary = {0.f, 0.f};
}
VD void trav_run(Trav &T, const DScene &S, uint32_t *stack, int stride, DCounters &cnt, int exit_below) {
    uint32_t nn = 0, nt = 0;
    for (;;) {
        bool act = T.cur != kTravDone;
        int n_act = __popcll(__ballot(act));
        if (n_act == 0 || n_act < exit_below) break;
        while (T.cur >= 0 && T.cur != kTravDone) {
            const float4 *q = reinterpret_cast<const float4 *>(S.nodes + T.cur);
            float4 n0 = q[0], n1 = q[1], n2 = q[2], n3 = q[3];
            ++nn;
            int32_t left = (int32_t) f2u(n3.x), right = (int32_t) f2u(n3.y);
            float tl, tr;
            bool hl = hit_box(f2v{n0.x, n0.y}, f2v{n0.z, n0.w}, f2v{n1.x, n1.y}, T.o, T.inv, T.best_t, &tl);
            bool hr = hit_box(f2v{n1.z, n1.w}, f2v{n2.x, n2.y}, f2v{n2.z, n2.w}, T.o, T.inv, T.best_t, &tr);
            if (hl && hr) {
                bool left_first = tl <= tr;
                if (T.sp < kStackDepth) { stack[T.sp * stride] = (uint32_t) (left_first ? right : left); ++T.sp; }
                T.cur = left_first ? left : right;
            } else if (hl) T.cur = left;
            else if (hr) T.cur = right;
            else if (T.sp > 0) { --T.sp; T.cur = (int32_t) stack[T.sp * stride]; }
            else T.cur = kTravDone;
        }
        if (T.cur != kTravDone) {
            uint32_t v = ~(uint32_t) T.cur;
            uint32_t first = v & kLeafFirstMask, count = (v >> 28) + 1u;
            bool stop = false;
            for (uint32_t i = 0; i < count; ++i) {
                const vmk_tri_pos *tp = S.tri_pos + first + i;
                float t, u, w;
                uint32_t inst, prim;
                ++nt;
                if (!intersect_tri(tp, T.o, T.d, &t, &u, &w, &inst, &prim)) continue;
                if (!(t > 0.f && t < T.t_max)) continue;
                if (T.any_hit) { T.found = true; stop = true; break; }
                bool better = !T.found ? (t <= T.best_t) : (t < T.best_t || (t == T.best_t && (inst < T.hit.inst || (inst == T.hit.inst && prim < T.hit.prim))));
                if (better) { T.best_t = t; T.hit.inst = inst; T.hit.prim = prim; T.hit.tri = first + i; T.hit.bary = {u, w}; T.found = true; }
            }
            if (stop) T.cur = kTravDone;
            else if (T.sp > 0) { --T.sp; T.cur = (int32_t) stack[T.sp * stride]; }
            else T.cur = kTravDone;
        }
    }
    cnt.nodes += nn; cnt.tris += nt;
}

}// namespace vmkd
