// dbvh.h — wave-cooperative BVH4 traversal: 4 lanes per ray, 16 rays in flight per wave64.
// Replaces ocarina::Accel trace_closest / trace_occlusion (base/mgr/geometry.cpp:168-185; OptiX in the reference).
// Hit selection rule (shared with the oracle): valid hits 0 < t < t_max, smallest t wins, equal t resolved towards
// the smaller (inst, prim).  Triangle test: watertight (Woop et al.) on world-space vertices, IEEE float32, no contraction.
//
// Why quads.  With one ray per lane (the first design) the counters on classroom read: 12.9 of 64 lanes
// active per VALU op and the texture-addresser (TA) 73 % busy — a wave64 memory instruction occupies the TA for ~19
// cycles however few lanes are live, and a 64 B BVH2 node cost each lane 4 of them.  Here the 4 lanes of a quad own ONE
// ray: on an internal node lane q loads and tests child q (2 x 16 B per lane, one 128 B line per quad), the quad orders
// its hits with three DPP quad_perm reads and pushes the far ones onto the ray's stack in LDS; on a leaf lane q tests
// triangle q.  A wave serves its 64 rays through 16 quads that pull the next ray from an LDS list as soon as their ray
// retires, so lanes whose path is dead or short cost nothing.
#pragma once
#include "dpath.h"

namespace vmkd {

#ifndef VMK_QUAD_STACK
#define VMK_QUAD_STACK 64
#endif
constexpr int kQuadStack = VMK_QUAD_STACK; // stack entries per ray; a build whose worst-case need exceeds it is rejected
// Trees deeper than that (bathroom2's PLOC tree needs 94) keep the entries beyond the LDS stack in HBM (DScene::stack_overflow):
// every lane of the quad writes and reads ITS OWN copy of the quad's overflow entries ([entry][lane], 256 B coalesced per entry), so
// no value crosses lanes through global memory — the refs of the other three lanes come over DPP in the (cold) overflow branch.
constexpr int kStackOverflow = 192;      // entries per ray beyond kQuadStack
constexpr uint32_t kOverflowWaves = 8192; // waves of one grid the overflow area serves (persistent grids use <= 6144)
constexpr int32_t kTravDone = 0x7fffffff;

struct Hit { uint32_t inst, prim, tri; V2 bary; };

// one per wave, in LDS (6.25 KB): staged rays / returned hits by lane, the list of lanes that have a ray, 16 ray stacks
struct WaveScratch {
    float4 ray[64][2];              // in: {o.xyz, t_max} {d.xyz, any_hit}   out (same slot): {inst, prim, tri, found} {u, v, -, -}
    uint32_t stack[kQuadStack][16]; // [level][quad]: the 16 quads of a wave hit 16 different banks
    uint32_t list[64];              // lanes with an active ray, compacted
};

// While a vertex is shaded no traversal is in flight and the wave's scratch is idle: the material code parks the lobe it
// hands to the out-of-line lobe routines there (dpath.h stage_lobe).  This lane's slot, as a byte offset in LDS.
static_assert(sizeof(WaveScratch) >= kLobeLdsDwords * 64u * 4u, "the parked lobe must fit the wave's traversal scratch");
VD uint32_t lobe_lds_slot(WaveScratch *ws) {
    return (uint32_t) (uintptr_t) (VMK_AS3 WaveScratch *) ws + (threadIdx.x & 63u) * 4u;
}

// DPP quad permutes (full rate, no LDS traffic).  Every use sits in quad-uniform control flow.
template<int CTRL> VD int32_t quad_perm_i(int32_t v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template<int CTRL> VD float quad_perm_f(float v) { return u2f((uint32_t) __builtin_amdgcn_mov_dpp((int32_t) f2u(v), CTRL, 0xf, 0xf, true)); }
constexpr int kQuadXor1 = 0xB1; // quad_perm:[1,0,3,2]
constexpr int kQuadXor2 = 0x4E; // quad_perm:[2,3,0,1]
constexpr int kQuadXor3 = 0x1B; // quad_perm:[3,2,1,0]

// Watertight ray / triangle test (Woop, Benthin, Wald, JCGT 2013) in float32, bit-identical to the oracle's restatement: vertices
// translated to the ray origin, permuted so that the ray's dominant axis is z, sheared so that the ray runs along +z, three 2-D edge
// functions.  Each edge function is computed from the two sheared vertices of its edge alone (commutative products, one subtraction),
// so two triangles that share an edge see equal or exactly negated values and a ray cannot slip between them — what OptiX guarantees
// behind the reference's trace_closest (geometry.cpp:168-174).  Zeros are inside for either sign.  u weighs p1, v weighs p2.
// Per-ray constants of the test, computed once when a quad takes a ray: the dominant axis kz of d (first maximum of |d|; k0: kz == 0,
// k2: kz == 2) and the shear (Sx, Sy, Sz) = (d_x' / d_z', d_y' / d_z', 1 / d_z') in the permuted frame.  The traversal keeps these
// instead of d.  The two minor axes go to x and y in whichever order costs one select each (kz = 0 -> (y, z, x), 1 -> (x, z, y),
// 2 -> (x, y, z)): every quantity of the test is invariant under swapping them.
struct TriRay { V3 S; bool k0, k2; };
VD V3 tri_permute(V3 v, bool k0, bool k2) { return mk3(k0 ? v.y : v.x, k2 ? v.y : v.z, k2 ? v.z : (k0 ? v.x : v.y)); }
VD TriRay tri_ray_setup(V3 d) {
    const float ax = abs_(d.x), ay = abs_(d.y), az = abs_(d.z);
    const bool k1 = ay > ax;
    TriRay r;
    r.k2 = az > (k1 ? ay : ax);
    r.k0 = !k1 && !r.k2;
    const V3 dp = tri_permute(d, r.k0, r.k2);
    r.S.z = 1.f / dp.z; r.S.x = dp.x * r.S.z; r.S.y = dp.y * r.S.z;
    return r;
}
VD bool intersect_tri(const vmk_tri_pos *tris, uint32_t index, V3 o, const TriRay &R, float *t_out, float *u_out, float *v_out, uint32_t *inst_out, uint32_t *prim_out) {
    // 48 B record as three 16 B loads at "scalar base + 32-bit offset" (vmk_build_accel checks n_tris * 48 < 4 GiB): no 64-bit address
    // arithmetic and no address register pair to keep (it used to be spilled and reloaded from scratch in front of every leaf)
    const uint32_t off = index * (uint32_t) sizeof(vmk_tri_pos);
    float4 a = ldg_off(tris, off), b = ldg_off(tris, off + 16u), c = ldg_off(tris, off + 32u);
    V3 p0 = {a.x, a.y, a.z}, p1 = {a.w, b.x, b.y}, p2 = {b.z, b.w, c.x};
    *inst_out = f2u(c.y); *prim_out = f2u(c.z);
    const V3 A = tri_permute(p0 - o, R.k0, R.k2), B = tri_permute(p1 - o, R.k0, R.k2), C = tri_permute(p2 - o, R.k0, R.k2);
    // (packed fp32: the shear of a vertex and the two products of an edge function are one v_pk_mul_f32 / v_pk_add_f32 each — the same
    // IEEE operations in the same order as the scalar form the oracle runs, so the bits do not change)
    const f2v Sxy = {R.S.x, R.S.y};
    const f2v Axy = f2v{A.x, A.y} - Sxy * A.z, Bxy = f2v{B.x, B.y} - Sxy * B.z, Cxy = f2v{C.x, C.y} - Sxy * C.z;
    const f2v pu = Cxy * f2v{Bxy.y, Bxy.x}, pv = Axy * f2v{Cxy.y, Cxy.x}, pw = Bxy * f2v{Axy.y, Axy.x};
    const float U = pu.x - pu.y, V = pv.x - pv.y, W = pw.x - pw.y; // Cx By - Cy Bx,  Ax Cy - Ay Cx,  Bx Ay - By Ax
    // (v_min3 / v_max3: a NaN operand is ignored here and propagated by the oracle's compare-and-select — either way a NaN edge function
    // ends in a NaN det and a NaN t, which no caller accepts)
    const float lo = __builtin_fminf(__builtin_fminf(U, V), W), hi = __builtin_fmaxf(__builtin_fmaxf(U, V), W);
    const bool inside = !(lo < 0.f && hi > 0.f); // the edge functions agree in sign (zeros are inside for either sign)
    const float det = U + V + W;
    // t from the triangle's plane in the unsheared frame (see the oracle's note): interpolated sheared depths are off by eps * |vertex - o|,
    // more than a spawned ray's offset on a large quad; d' . N = d_z * det, so one reciprocal serves t, u and v.
    // Straight-line code: the oracle's early returns (outside, det == 0, t out of range) are folded into ONE predicate — a det of 0 makes
    // inv infinite and t infinite or NaN, which fails the range test below like everything else that is not a hit.
    const V3 e1 = B - A, e2 = C - A;
    const V3 N = cross(e2, e1); // this orientation has N . (Sx, Sy, 1) = U + V + W
    const float inv = 1.f / det;
    const float t = dot(A, N) * R.S.z * inv;
    // the hit lies within the triangle's extent along the dominant axis (padded by 2^-14): filters the arbitrary t of a triangle seen
    // edge-on, which would otherwise be reported or not depending on the order the leaves are visited in (see the oracle's note)
    const float zlo = __builtin_fminf(__builtin_fminf(A.z, B.z), C.z) * R.S.z, zhi = __builtin_fmaxf(__builtin_fmaxf(A.z, B.z), C.z) * R.S.z;
    const float tlo = zlo < zhi ? zlo : zhi, thi = zlo < zhi ? zhi : zlo;
    const float pad = 1.f / 16384.f;
    *t_out = t; *u_out = V * inv; *v_out = W * inv;
    return inside && det != 0.f && t * (1.f + pad) >= tlo && t * (1.f - pad) <= thi;
}

// conservative slab test; NaN slabs (0 * inf) are ignored by the min/max (IEEE minNum/maxNum on v_min/v_max_f32).
// The node stores each slab as a (min, max) pair so that both planes of a slab go through one v_pk_add_f32 + one
// v_pk_mul_f32 (gfx950 packed fp32).  Measured and rejected: t = fma(b, inv, -o*inv) is not conservative for rays that
// start on a box face (every bounce ray does): its absolute error |o*inv|*2^-24 is unbounded relative to (b-o)*inv.
// The test is padded by 1e-4 relative on both ends.
VD bool hit_box(f2v bx, f2v by, f2v bz, V3 o, V3 inv, float t_far, float *t_near_out) {
    f2v tx = (bx - o.x) * inv.x, ty = (by - o.y) * inv.y, tz = (bz - o.z) * inv.z;
    float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)), __builtin_fmaxf(__builtin_fminf(tz.x, tz.y), 0.f));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)), __builtin_fminf(__builtin_fmaxf(tz.x, tz.y), t_far));
    *t_near_out = tn;
    // The margin also decides whether a leaf whose triangle computes a t slightly OUTSIDE its own box (ill-conditioned Moeller-Trumbore,
    // typically two triangles meeting at an edge) is still visited once the culling bound has tightened to a near-equal hit: with 1e-6
    // that depended on when the bound tightened — on the wave's scheduling — and two runs of the headline launch differed in about one
    // path per 5e8; with 1e-4 four runs of 5.3e8 paths each agree in every bit (tools/gpu_determinism.py), at no measurable cost.
#ifndef VMK_CULL_LO
#define VMK_CULL_LO 0.9999f
#define VMK_CULL_HI 1.0001f
#endif
    return tn <= tf * (VMK_CULL_HI / VMK_CULL_LO); // (one multiply: the two factors folded)
}

VD void wave_lds_fence() { // LDS written by other lanes of this wave is visible after this point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- where the rays come from and where the hits go ----------------------------------------------------------
// Megakernel: every lane stages its own ray in LDS, the quads work through the compacted list of staged lanes.
struct LdsRayIO {
    WaveScratch *ws;
    uint32_t n, next; // wave-uniform
    VD bool more() const { return next < n; }
    VD uint32_t claim(uint32_t count) { uint32_t b = next; next = min(next + count, n); return b; }
    VD bool load(uint32_t idx, int &owner, V3 &o, V3 &d, float &t_max, bool &anyh) {
        if (idx >= n) return false;
        owner = (int) ws->list[idx];
        float4 a = ws->ray[owner][0], b = ws->ray[owner][1];
        o = mk3(a.x, a.y, a.z); t_max = a.w; d = mk3(b.x, b.y, b.z); anyh = b.w != 0.f;
        return true;
    }
    VD void store(int owner, bool found, uint32_t inst, uint32_t prim, uint32_t tri, float u, float v) { // one lane per ray
        ws->ray[owner][0] = make_float4(u2f(inst), u2f(prim), u2f(tri), found ? 1.f : 0.f);
        ws->ray[owner][1] = make_float4(u, v, 0.f, 0.f);
    }
};
// Replay kernel: SoA ray planes in HBM shared by the whole grid.  A wave takes `chunk` rays from the pool with one atomic
// (a claim per refill batch serialises on the one counter: 2 M rays took 4.3 ms instead of 0.8) and hands them to its
// quads as they fall idle, so every quad stays busy until the pool is empty (no per-64-ray tail).
struct GlobalRayIO {
    uint32_t chunk; // rays per claim: large enough to keep the counter cold, small enough to balance the waves
    const float *org, *dir, *tmax;
    uint32_t n;
    uint32_t *queue;
    uint4 *hit_out;
    bool any_hit;
    uint32_t lo, hi; // wave-uniform: unclaimed part of the wave's current chunk
    bool exhausted;  // wave-uniform: the pool has no further chunk
    VD bool more() const { return lo < hi || !exhausted; }
    VD uint32_t claim(uint32_t count) { // called in wave-uniform control flow; rays [base, lo) are handed out
        if (lo >= hi) {
            uint32_t b = 0;
            if ((threadIdx.x & 63u) == 0) b = atomicAdd(queue, chunk);
            b = (uint32_t) __builtin_amdgcn_readfirstlane((int) b);
            lo = min(b, n); hi = min(b + chunk, n);
            if (b + chunk >= n) exhausted = true;
        }
        uint32_t base = lo;
        lo = min(lo + count, hi);
        return base;
    }
    VD bool load(uint32_t idx, int &owner, V3 &o, V3 &d, float &t_max, bool &anyh) {
        if (idx >= lo) return false; // beyond what this claim handed out: the quad asks again in the next batch
        owner = (int) idx;
        o = mk3(org[idx], org[n + idx], org[2 * n + idx]); d = mk3(dir[idx], dir[n + idx], dir[2 * n + idx]); t_max = tmax[idx]; anyh = any_hit;
        return true;
    }
    VD void store(int owner, bool found, uint32_t inst, uint32_t prim, uint32_t tri, float u, float v) {
        hit_out[owner] = any_hit ? make_uint4(found ? 1u : 0u, 0, 0, 0) : make_uint4(inst, prim, f2u(u), f2u(v));
    }
};

#ifndef VMK_NODE_QUADS_MIN
#define VMK_NODE_QUADS_MIN 8 // leave the node phase when fewer quads than this are on internal nodes and others wait
#endif
#ifndef VMK_REFILL_QUADS_MIN
#define VMK_REFILL_QUADS_MIN 4 // hand back hits / take new rays once this many quads are idle (or nothing else is left); 4: 339.9, 6: 336.5, 8: 341 ms (classroom, 32 spp)
#endif

// The traversal loop of one wave; EVERY lane of the wave must be here (convergently).
//
// Loop structure ("while-while" with one postponed leaf per ray): the wave alternates between a node phase, in which
// every quad that sits on an internal node takes steps, and a leaf phase.  A quad that reaches a leaf parks it in
// `pend` and keeps descending with the next stack entry, so the node phase only loses a quad when it holds two leaves;
// the price is that the parked leaf cannot tighten best_t for the nodes visited meanwhile.  Hits are handed back and new
// rays taken in batches (VMK_REFILL_QUADS_MIN) because that block costs as much as a node step for the whole wave.
// Returns the number of rays this quad-lane started (lane q == 0 only) in *n_rays for the callers' counters.
// COUNT: tally node fetches and triangle tests (cnt.nodes / cnt.tris, the algorithmic-bytes side of the roofline); the two adds sit
// in the innermost loops, so timed launches can run the COUNT = false instance and take the tallies from a sibling launch over
// the same (deterministic) rays.
// DEEP: the tree's worst-case stack need exceeds the LDS stack — entries beyond it live in HBM (see kStackOverflow).  A compile-time
// variant: the test on every push / pop costs the hot kernel 8 % on classroom when it is there unconditionally.
// ANYHIT: every ray of the call is an occlusion query (the shadow traversal of path_bounce, k_trace with any_hit).  A compile-time
// variant because such rays need none of the closest-hit machinery and the loop is VALU-issue bound: the children of a node are not
// ranked by distance (any hit ends the ray: the quad's hit mask from one ballot places the pushes), no candidate (t, u, v, inst, prim)
// is kept or merged, and the culling bound never tightens.
template<class IO, bool COUNT = true, bool DEEP = false, bool ANYHIT = false>
VD void traverse_core(const DScene &S, IO &io, WaveScratch *ws, DCounters &cnt, uint32_t *n_rays) {
    const uint32_t lane = threadIdx.x & 63u, q = lane & 3u, quad = lane >> 2;
    // ---- per-quad traversal state, replicated in the quad's 4 lanes ----
    int32_t cur = kTravDone;         // internal node (0 <= cur < kEmptyRef), leaf (< 0) or kTravDone
    int32_t pend = kTravDone;        // parked leaf or kTravDone
    int sp = 0;
    int owner = -1;
    V3 o = mk3(0.f), inv = mk3(0.f);
    TriRay tray = {mk3(0.f), false, false}; // the watertight triangle test's view of the ray direction (kept instead of d)
    uint32_t oct4 = 0;                      // 4 x the sign octant of d (bit 0: d.x < 0, bit 1: d.y < 0, bit 2: d.z < 0): selects the child-order nibble
    float t_max = 0.f, best_t = 0.f; // best_t: quad-wide culling bound
    bool anyh = false;
    // lane-local best candidate (merged across the quad when the ray retires)
    float bt = 0.f, bu = 0.f, bv = 0.f;
    uint32_t binst = VMK_INVALID, bprim = VMK_INVALID, btri = VMK_INVALID;
    bool found = false;
    uint32_t nn = 0, nt = 0, nr = 0;
    // this wave's slice of the HBM stack overflow (null unless the tree needs it)
    // Deep trees: the entries beyond the LDS stack live in HBM.  The address of this lane's slice is NOT kept in registers — it is rebuilt
    // from scalars inside the (cold) branches that touch it, so a tree that fits the LDS stack by a wide margin most of the time pays one
    // compare per pop and per push for the variant.
    const bool has_ovf = DEEP && S.stack_overflow != nullptr;
    auto ovf_slot = [&](int entry) -> uint32_t * {
        return S.stack_overflow + (((size_t) (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * kStackOverflow + (size_t) (entry - kQuadStack)) * 64u + lane);
    };
#define VMK_POP() do { if (sp > 0) { --sp; if (DEEP && __builtin_expect(sp >= kQuadStack, 0)) cur = (int32_t) *ovf_slot(sp); else cur = (int32_t) ws->stack[sp][quad]; } else cur = kTravDone; } while (0)

    // Loop shape: the OUTER loop is one hand-back / refill round (it also starts the traversal: every quad is idle and owns nothing); the
    // INNER do-while alternates node and leaf phases and has a single back-edge.  With the refill inside the same loop as the phases (two
    // `continue`s and a fall-through back-edge) the register allocator copied the whole per-ray state — some 25 registers — at the merge
    // on every round whether or not a quad took a ray; keeping the state's redefinition out of the inner loop confines that to the
    // refill rounds.
    for (;;) {
        // ================= hand back hits, take new rays =================
        {
            const bool idle = cur == kTravDone && pend == kTravDone;
            if constexpr (ANYHIT) { if (idle && owner >= 0) { if (q == 0) io.store(owner, found, VMK_INVALID, VMK_INVALID, VMK_INVALID, 0.f, 0.f); owner = -1; } }
            else if (idle && owner >= 0) {
                // ---- retire: the lane that holds the quad's best candidate (min t, then inst, then prim) returns it ----
                uint32_t mb = found ? f2u(bt) : 0x7f800000u; // (+inf; bit-pattern minimum, see the leaf phase)
                mb = min(mb, (uint32_t) quad_perm_i<kQuadXor1>((int32_t) mb));
                mb = min(mb, (uint32_t) quad_perm_i<kQuadXor2>((int32_t) mb));
                const float m = u2f(mb);
                const bool c1 = found && bt == m;
                uint32_t ki = c1 ? binst : 0xffffffffu;
                ki = min(ki, (uint32_t) quad_perm_i<kQuadXor1>((int32_t) ki));
                ki = min(ki, (uint32_t) quad_perm_i<kQuadXor2>((int32_t) ki));
                const bool c2 = c1 && binst == ki;
                uint32_t kp = c2 ? bprim : 0xffffffffu;
                kp = min(kp, (uint32_t) quad_perm_i<kQuadXor1>((int32_t) kp));
                kp = min(kp, (uint32_t) quad_perm_i<kQuadXor2>((int32_t) kp));
                const bool none = !(m < __builtin_inff());
                if (c2 && bprim == kp) io.store(owner, true, binst, bprim, btri, bu, bv);
                else if (none && q == 0) io.store(owner, false, VMK_INVALID, VMK_INVALID, VMK_INVALID, 0.f, 0.f);
                owner = -1;
            }
            // ---- refill ----
            const unsigned long long idle_mask = __ballot(idle && q == 0);
            const uint32_t base = io.more() ? io.claim((uint32_t) __popcll(idle_mask)) : 0xffffffffu;
            if (idle && base != 0xffffffffu) {
                uint32_t idx = base + (uint32_t) __popcll(idle_mask & ((1ull << (lane & ~3u)) - 1ull));
                V3 d;
                if (io.load(idx, owner, o, d, t_max, anyh)) {
                    tray = tri_ray_setup(d);
                    oct4 = ((f2u(d.x) >> 31) | ((f2u(d.y) >> 31) << 1) | ((f2u(d.z) >> 31) << 2)) * 4u;
                    // v_rcp_f32 (1 ulp) is enough here: inv only feeds the padded, conservative slab test
                    inv = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)};
                    best_t = t_max; cur = S.root; sp = 0;
                    found = false; bt = t_max; binst = VMK_INVALID; bprim = VMK_INVALID; btri = VMK_INVALID; bu = 0.f; bv = 0.f;
                    nr += q == 0 ? 1u : 0u;
                }
            }
            if (!__any(owner >= 0)) break; // nothing in flight and nothing left to take
        }
        bool again;
        do {
        // ================= node phase =================
        for (bool first = true;; first = false) {
            const bool at_node = (uint32_t) cur < (uint32_t) kEmptyRef;
            const unsigned long long nmask = __ballot(at_node);
            if (nmask == 0) break;
            if (!first && __popcll(nmask) < 4 * VMK_NODE_QUADS_MIN) {
                const bool waiting = !at_node && (pend != kTravDone || cur < 0 || owner >= 0 || io.more());
                if (__any(waiting)) break;
            }
            if (at_node) { // lane q tests child q
                const uint32_t off = (uint32_t) cur * (uint32_t) sizeof(BvhNode) + q * 32u; // < 4 GiB: vmk_build_accel checks
                float4 a = ldg_off(S.nodes, off), b = ldg_off(S.nodes, off + 16u);
                int32_t ref = (int32_t) f2u(b.z);
                float tn;
                // (unused slots hold a far-away point box; the explicit ref test keeps NaN rays, whose slab test passes
                // everywhere, from walking into them)
                bool h = hit_box(f2v{a.x, a.y}, f2v{a.z, a.w}, f2v{b.x, b.y}, o, inv, best_t, &tn) && ref != kEmptyRef;
                if constexpr (COUNT) nn += q == 0 ? 1u : 0u;
                int rank, n;
                int32_t cand;
                uint32_t key = 0, k1 = 0, k2 = 0, k3 = 0;
                if constexpr (ANYHIT) {
                    // no order among the children: the quad's 4-bit hit mask (one ballot) gives the count, the child to descend into (the
                    // highest hit lane) and every other hit lane's place on the stack
                    const uint32_t hits = (uint32_t) (__ballot(h) >> (lane & ~3u)) & 0xfu;
                    n = __popc(hits);
                    const int top = 31 - __clz((int) hits);         // -1 when no child is hit
                    rank = (int) q == top ? 0 : 1 + __popc(hits & ((1u << q) - 1u)); // 1 .. n-1 for the others, in lane order
                    cand = (h && rank == 0) ? ref : (int32_t) 0x80000000;
                    cand = max(cand, quad_perm_i<kQuadXor1>(cand));
                    cand = max(cand, quad_perm_i<kQuadXor2>(cand));
                } else {
                // Order of the children: NOT sorted by entry distance at run time.  The node carries, per child and per sign octant of the
                // ray direction, the set of siblings that precede the child front to back (k_bvh4_level); intersected with the quad's hit
                // mask (one ballot) that is the child's rank among the children that were hit.  The order only steers culling, it never
                // changes which hit is returned.  (Exact distance ranking — a key per lane, three DPP reads and compares — cost twice
                // the instructions in a loop that is bound by VALU issue.)
                const uint32_t hits = (uint32_t) (__ballot(h) >> (lane & ~3u)) & 0xfu;
                const uint32_t before = (f2u(b.w) >> oct4) & 0xfu;
                n = __popc(hits);
                rank = __popc(hits & before);
                cand = (h && rank == 0) ? ref : (int32_t) 0x80000000;
                cand = max(cand, quad_perm_i<kQuadXor1>(cand));
                cand = max(cand, quad_perm_i<kQuadXor2>(cand));
                }
                if (h && rank > 0) { // far children: the nearest of them ends up on top
                    int slot = sp + (n - 1 - rank);
                    if (slot < kQuadStack) ws->stack[slot][quad] = (uint32_t) ref;
                }
                if constexpr (DEEP) if (__builtin_expect(sp + n - 1 > kQuadStack, 0) && has_ovf) { // cold: some of the far children land beyond the LDS stack (deep trees only)
                    { // the same placement expressed as keys (rank, then lane)
                        key = h ? (uint32_t) rank * 4u + q : 0xffffffffu;
                        k1 = (uint32_t) quad_perm_i<kQuadXor1>((int32_t) key); k2 = (uint32_t) quad_perm_i<kQuadXor2>((int32_t) key); k3 = (uint32_t) quad_perm_i<kQuadXor3>((int32_t) key);
                    }
                    const int32_t r1 = quad_perm_i<kQuadXor1>(ref), r2 = quad_perm_i<kQuadXor2>(ref), r3 = quad_perm_i<kQuadXor3>(ref);
                    const uint32_t kk[4] = {key, k1, k2, k3};
                    const int32_t rr[4] = {ref, r1, r2, r3};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { // every member of the quad, as seen from this lane
                        if (kk[j] == 0xffffffffu) continue;
                        int rank_j = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) rank_j += kk[i] < kk[j] ? 1 : 0;
                        const int slot_j = sp + (n - 1 - rank_j);
                        if (rank_j > 0 && slot_j >= kQuadStack && slot_j < kQuadStack + kStackOverflow) *ovf_slot(slot_j) = (uint32_t) rr[j];
                    }
                }
                // The pops below read entries that OTHER lanes of the quad have just written.  The hardware keeps the LDS
                // operations of a wave in issue order, so all that is needed is that every lane's store is ISSUED before
                // any lane's load: a convergent fence between the two, which the compiler may neither duplicate into the
                // divergent `if` above nor sink loads across (no instruction is emitted for it).  Without it the order of
                // the two was a property of the block layout the optimiser happened to pick ("UB").
                wave_lds_fence();
                if (n > 0) { cur = cand; sp = min(sp + n - 1, has_ovf ? kQuadStack + kStackOverflow : kQuadStack); }
                else VMK_POP();
                if (cur < 0 && pend == kTravDone) { pend = cur; VMK_POP(); } // park the leaf, keep descending
            }
        }
        // ================= leaf phase: lane q tests triangle q of the parked leaf =================
        if (pend != kTravDone) {
            uint32_t v = ~(uint32_t) pend;
            uint32_t first = v & kLeafFirstMask, count = (v >> 28) + 1u;
            if (q < count) {
                float t, u, w;
                uint32_t inst, prim;
                if constexpr (COUNT) ++nt;
                if (intersect_tri(S.tri_pos, first + q, o, tray, &t, &u, &w, &inst, &prim) && t > 0.f && t < t_max) {
                    if constexpr (ANYHIT) found = true;
                    else {
                        bool better = !found || t < bt || (t == bt && (inst < binst || (inst == binst && prim < bprim)));
                        if (better) { found = true; bt = t; binst = inst; bprim = prim; btri = first + q; bu = u; bv = w; }
                    }
                }
            }
            pend = kTravDone;
            if constexpr (ANYHIT) {
                // occlusion query: one hit in the quad ends the ray (the flag is quad-uniform from here on)
                found = ((uint32_t) (__ballot(found) >> (lane & ~3u)) & 0xfu) != 0u;
                if (found) cur = kTravDone;
            } else {
                // (quad minimum over the BIT PATTERNS: distances are positive, so unsigned integer order is float order, and v_min_u32
                // takes its DPP operand directly — a float min would spend a move and two canonicalising max per step on top)
                uint32_t mb = f2u(found ? bt : t_max);
                mb = min(mb, (uint32_t) quad_perm_i<kQuadXor1>((int32_t) mb));
                mb = min(mb, (uint32_t) quad_perm_i<kQuadXor2>((int32_t) mb));
                const float m = u2f(mb);
                best_t = m;
                if (anyh && m < t_max) cur = kTravDone; // (an occlusion query in a mixed pool: some lane of the quad has a hit)
            }
        }
        if (cur < 0) { pend = cur; VMK_POP(); } // a second leaf was waiting: park it for the next leaf phase
        // ---- back to the refill round once enough quads are idle (or nothing else is left) ----
        const bool idle = cur == kTravDone && pend == kTravDone;
        const bool want = idle && (owner >= 0 || io.more());
        const unsigned long long want_mask = __ballot(want);
        const bool any_busy = __any(!idle);
        again = any_busy && (want_mask == 0 || __popcll(want_mask) < 4 * VMK_REFILL_QUADS_MIN);
        } while (again);
    }
#undef VMK_POP
    cnt.nodes += nn; cnt.tris += nt;
    if (n_rays) *n_rays = nr;
}

// Trace the rays of all lanes of the wave.  EVERY lane of the wave must call this (convergently); `active` says whether
// the lane has a ray.  any_hit rays stop at the first valid hit (occlusion query), the others return the closest hit.
// Returns found; `hit` is filled for closest-hit rays.
template<bool COUNT = true, bool DEEP = false, bool ANYHIT = false>
VD bool traverse_wave(const DScene &S, const Ray &r, bool active, WaveScratch *ws, Hit &hit, DCounters &cnt) {
    constexpr bool any_hit = ANYHIT;
    const uint32_t lane = threadIdx.x & 63u;
    hit.inst = VMK_INVALID; hit.prim = VMK_INVALID; hit.tri = VMK_INVALID; hit.bary = {0.f, 0.f};
    if (S.n_tris == 0) return false;
    // ---- stage the rays of the active lanes ----
    const unsigned long long act_mask = __ballot(active);
    const uint32_t n_act = (uint32_t) __popcll(act_mask);
    if (n_act == 0) return false;
    if (active) {
        ws->ray[lane][0] = make_float4(r.o.x, r.o.y, r.o.z, r.t_max);
        ws->ray[lane][1] = make_float4(r.d.x, r.d.y, r.d.z, any_hit ? 1.f : 0.f);
        ws->list[__popcll(act_mask & ((1ull << lane) - 1ull))] = lane;
    }
    wave_lds_fence();
    LdsRayIO io = {ws, n_act, 0};
    traverse_core<LdsRayIO, COUNT, DEEP, ANYHIT>(S, io, ws, cnt, nullptr);
    wave_lds_fence();
    bool res = false;
    if (active) {
        float4 a = ws->ray[lane][0], b = ws->ray[lane][1];
        res = a.w != 0.f;
        if (res) { hit.inst = f2u(a.x); hit.prim = f2u(a.y); hit.tri = f2u(a.z); hit.bary = {b.x, b.y}; }
    }
    wave_lds_fence(); // the scratch is restaged by the next call
    return res;
}

}// namespace vmkd
