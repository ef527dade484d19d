// vmk.hip — libvmk.so: the gfx950 path-tracing backend behind include/vmk.h.
//
// Kernels (all hand-written HIP, wave64):
//   k_render         the megakernel: ray-gen -> wave-cooperative BVH4 traversal (dbvh.h) -> polymorphic BSDF eval/sample ->
//                    NEE -> Russian roulette -> in-register film accumulation; persistent lanes pull (pixel) work items
//                    with a wavefront ballot so finished lanes are refilled instead of idling.
//   k_morton / k_ploc_* / k_bvh4_level   GPU BVH build (63-bit Morton codes -> radix sort -> PLOC merge rounds ->
//                    depth-first triangle order -> top-down collapse into 128 B BVH4 nodes)
//   k_trace          traversal-only replay over SoA ray buffers (roofline measurement, parity of hits)
//   k_tonemap        exposure / tone map / gamma epilogue
//   k_test           per-function device unit entry points for the parity tests
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h> // types only: the library is loaded with dlopen on first use (vmk_comm_*)
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "drender.h"

using namespace vmkd;

// the hero-spectrum instances of the megakernel live in vmk_hero.hip (3 wavelengths per path) and vmk_hero4.hip (4)
#define VMK_HERO_DECL(prefix) \
hipError_t prefix##occupancy(bool full, bool media, bool count, bool deep, int *blocks_per_cu); \
hipError_t prefix##launch_unit_path(hipStream_t stream, const void *scene, const void *params, uint32_t n, const float *in, uint32_t in_stride, float *out, uint32_t out_stride); \
hipError_t prefix##launch_render(bool full, bool media, bool count, bool deep, unsigned blocks, hipStream_t stream, const void *rest, size_t rest_bytes, const void *scene, size_t scene_bytes); \
hipError_t prefix##launch_aov(unsigned blocks, hipStream_t stream, const void *args, size_t args_bytes);
VMK_HERO_DECL(vmk_hero_)
VMK_HERO_DECL(vmk_hero4_)

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) { ctx->error = std::string(#expr) + ": " + hipGetErrorString(e_); return VMK_ERR_HIP; } \
    } while (0)

constexpr int kDefaultTile = 32;

// ---- tile ownership (include/vmk.h vmk_tiles) ----
static uint32_t gcd_u32(uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; }
extern "C" uint32_t vmk_tile_skew(uint32_t world) {
    if (world <= 2) return 1;
    uint32_t s = (uint32_t) (0.38 * (double) world + 0.999999);
    if (!(s & 1u)) ++s;
    while (gcd_u32(s, world) != 1) s += 2;
    return s;
}
static std::vector<uint32_t> owned_tiles(uint32_t tiles_x, uint32_t tiles_y, uint32_t rank, uint32_t world) {
    std::vector<uint32_t> t;
    const uint32_t skew = vmk_tile_skew(world);
    for (uint32_t ty = 0; ty < tiles_y; ++ty)
        for (uint32_t tx = 0; tx < tiles_x; ++tx)
            if ((tx + (uint64_t) skew * ty) % world == rank) t.push_back(ty * tiles_x + tx);
    return t;
}

// ---------------------------------------------------------------------------------------------------------
// LBVH build
// ---------------------------------------------------------------------------------------------------------
// 63-bit Morton code: 21 bits per axis.  (Scenes like classroom carry a 220 m backdrop around a 10 m room: with 10-bit
// cells most triangles share a code and the hierarchy degenerates to index order — 69 node visits per ray.)
__device__ __forceinline__ uint64_t expand_bits21(uint64_t v) {
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__global__ void k_morton(const vmk_tri_pos *tris, uint32_t n, float3 bmin, float3 inv_ext, uint64_t *keys, uint32_t *vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const vmk_tri_pos &t = tris[i];
    float c[3];
    for (int a = 0; a < 3; ++a) {
        float lo = fminf(t.p0[a], fminf(t.p1[a], t.p2[a])), hi = fmaxf(t.p0[a], fmaxf(t.p1[a], t.p2[a]));
        c[a] = 0.5f * (lo + hi);
    }
    const float S = 2097152.f; // 2^21
    float x = fminf(fmaxf((c[0] - bmin.x) * inv_ext.x * S, 0.f), S - 1.f);
    float y = fminf(fmaxf((c[1] - bmin.y) * inv_ext.y * S, 0.f), S - 1.f);
    float z = fminf(fmaxf((c[2] - bmin.z) * inv_ext.z * S, 0.f), S - 1.f);
    keys[i] = (expand_bits21((uint64_t) x) << 2) | (expand_bits21((uint64_t) y) << 1) | expand_bits21((uint64_t) z);
    vals[i] = i;
}
__global__ void k_reorder(const uint32_t *vals, uint32_t n, const vmk_tri_pos *pos_in, const vmk_tri_attr *attr_in,
                          vmk_tri_pos *pos_out, vmk_tri_attr *attr_out, uint32_t *lookup) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t src = vals[i];
    pos_out[i] = pos_in[src];
    attr_out[i] = attr_in[src];
    lookup[src] = i;
}

// ---------------------------------------------------------------------------------------------------------
// PLOC hierarchy (Meister & Bittner 2018, "Parallel Locally-Ordered Clustering for BVH Construction") on the
// Morton-sorted triangles: every round each cluster finds its nearest neighbour (smallest merged surface area) within
// +-radius positions, mutual pairs merge.  Measured on classroom: 19-21 node visits per closest-hit ray against 47 for
// the plain Karras LBVH over the same order (the scene has a 220 m backdrop around a 10 m room), on par with binned SAH.
// Node ids: [0, n) leaves (sorted-triangle index), [n, 2n-1) internal in creation order (children < parent).
// ---------------------------------------------------------------------------------------------------------
struct Box6 { float lo[3], hi[3]; };
__device__ __forceinline__ float box_union_area(const Box6 &a, const Box6 &b) {
    float d0 = fmaxf(a.hi[0], b.hi[0]) - fminf(a.lo[0], b.lo[0]);
    float d1 = fmaxf(a.hi[1], b.hi[1]) - fminf(a.lo[1], b.lo[1]);
    float d2 = fmaxf(a.hi[2], b.hi[2]) - fminf(a.lo[2], b.lo[2]);
    return d0 * d1 + d1 * d2 + d0 * d2;
}
__global__ void k_ploc_init(const vmk_tri_pos *tris, int n, Box6 *box, int *cl, int *parent, int *count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const vmk_tri_pos &t = tris[i];
    Box6 b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(t.p0[a], fminf(t.p1[a], t.p2[a])); b.hi[a] = fmaxf(t.p0[a], fmaxf(t.p1[a], t.p2[a])); }
    box[i] = b; cl[i] = i; parent[i] = -1; count[i] = 1;
}
__global__ void k_ploc_nn(int m, const int *cl, const Box6 *box, int radius, int *nn) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Box6 me = box[cl[i]];
    float best = 3.0e38f; int bj = -1;
    int j0 = max(0, i - radius), j1 = min(m - 1, i + radius);
    for (int j = j0; j <= j1; ++j) {
        if (j == i) continue;
        float a = box_union_area(me, box[cl[j]]);
        if (a < best) { best = a; bj = j; }
    }
    nn[i] = bj;
}
// flags: low 32 bits = cluster survives into the next round, high 32 bits = this position creates a node
__global__ void k_ploc_flags(int m, const int *nn, unsigned long long *flags, int force_pairs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    int j = force_pairs ? ((i ^ 1) < m ? (i ^ 1) : -1) : nn[i];
    bool mutual = j >= 0 && (force_pairs ? true : nn[j] == i);
    bool merge = mutual && i < j, drop = mutual && i > j;
    flags[i] = ((unsigned long long) (merge ? 1u : 0u) << 32) | (unsigned long long) (drop ? 0u : 1u);
}
__global__ void k_ploc_merge(int m, const int *cl, const int *nn, const unsigned long long *flags, const unsigned long long *scan, int node_base,
                             Box6 *box, int *left, int *right, int *parent, int *count, int *cl_out, int force_pairs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    unsigned long long f = flags[i], sc = scan[i];
    if ((f & 0xffffffffull) == 0) return;
    int pos = (int) (sc & 0xffffffffull);
    if (f >> 32) {
        int j = force_pairs ? (i ^ 1) : nn[i];
        int id = node_base + (int) (sc >> 32);
        int a = cl[i], b = cl[j];
        Box6 ba = box[a], bb = box[b], u;
        for (int k = 0; k < 3; ++k) { u.lo[k] = fminf(ba.lo[k], bb.lo[k]); u.hi[k] = fmaxf(ba.hi[k], bb.hi[k]); }
        box[id] = u; left[id] = a; right[id] = b; parent[a] = id; parent[b] = id; parent[id] = -1; count[id] = count[a] + count[b];
        cl_out[pos] = id;
    } else cl_out[pos] = cl[i];
}
// depth-first triangle order: position of a node's first leaf = sum of the left-sibling sizes on its root path
__device__ __forceinline__ int ploc_first(int c, const int *left, const int *right, const int *parent, const int *count) {
    int pos = 0;
    for (int p = parent[c]; p >= 0; c = p, p = parent[p])
        if (right[p] == c) pos += count[left[p]];
    return pos;
}
__global__ void k_ploc_place(int n, const int *left, const int *right, const int *parent, const int *count, const uint32_t *orig,
                             const vmk_tri_pos *pos_in, const vmk_tri_attr *attr_in, vmk_tri_pos *pos_out, vmk_tri_attr *attr_out,
                             uint32_t *lookup) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int p = ploc_first(i, left, right, parent, count);
    pos_out[p] = pos_in[i]; attr_out[p] = attr_in[i];
    lookup[orig[i]] = (uint32_t) p;
}
// ---------------------------------------------------------------------------------------------------------
// BVH2 -> BVH4 collapse, top-down by levels.  A work item is a binary node that becomes a 4-wide node: its two
// children are opened (largest surface area first) until four slots are filled or only leaves remain; subtrees of
// <= kMaxLeafTris triangles become leaves (their triangles are contiguous in the depth-first order, and slots keep the
// left-to-right order so `first` stays the running sum of the left siblings).  Internal children get their output
// index from an atomic counter and are queued for the next level.  `acc` carries the stack entries already pending on
// the path from the root, so max(acc + children - 1) is the exact worst-case traversal stack need.
// ---------------------------------------------------------------------------------------------------------
struct Bvh4Work { int id, out, first, acc; };
__device__ __forceinline__ float box_area(const Box6 &b) {
    float d0 = b.hi[0] - b.lo[0], d1 = b.hi[1] - b.lo[1], d2 = b.hi[2] - b.lo[2];
    return d0 * d1 + d1 * d2 + d0 * d2;
}
__global__ void k_bvh4_level(const Bvh4Work *in, int n_in, Bvh4Work *out_q, int *out_count, int *node_counter, const Box6 *box,
                             const int *left, const int *right, const int *count, BvhNode *nodes, uint32_t *n_leaves, int *max_need) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    Bvh4Work w = in[i];
    int slot[4], first[4], ns = 2;
    slot[0] = left[w.id]; slot[1] = right[w.id];
    first[0] = w.first; first[1] = w.first + count[slot[0]];
    while (ns < 4) {
        int pick = -1; float best = -1.f;
        for (int k = 0; k < ns; ++k) {
            if (count[slot[k]] <= kMaxLeafTris) continue;
            float a = box_area(box[slot[k]]);
            if (a > best) { best = a; pick = k; }
        }
        if (pick < 0) break;
        for (int k = ns; k > pick + 1; --k) { slot[k] = slot[k - 1]; first[k] = first[k - 1]; }
        int c = slot[pick];
        slot[pick] = left[c]; slot[pick + 1] = right[c];
        first[pick + 1] = first[pick] + count[left[c]];
        ++ns;
    }
    BvhNode nd;
    uint32_t leaves = 0;
    for (int k = 0; k < 4; ++k) {
        BvhChild &ch = nd.child[k];
        ch.pad = 0;
        if (k >= ns) { ch.x[0] = ch.y[0] = ch.z[0] = 3.0e38f; ch.x[1] = ch.y[1] = ch.z[1] = 3.0e38f; ch.ref = kEmptyRef; continue; } // unused slot: a point no ray reaches
        Box6 b = box[slot[k]];
        ch.x[0] = b.lo[0]; ch.x[1] = b.hi[0]; ch.y[0] = b.lo[1]; ch.y[1] = b.hi[1]; ch.z[0] = b.lo[2]; ch.z[1] = b.hi[2];
        int c = count[slot[k]];
        if (c <= kMaxLeafTris) { ch.ref = (int32_t) ~(((uint32_t) first[k] & kLeafFirstMask) | ((uint32_t) (c - 1) << 28)); ++leaves; }
        else {
            int o = atomicAdd(node_counter, 1);
            ch.ref = o;
            out_q[atomicAdd(out_count, 1)] = {slot[k], o, first[k], w.acc + ns - 1};
        }
    }
    // Front-to-back order without a sort at run time: child q's spare word holds, for each of the 8 sign octants of a ray direction, the
    // 4-bit set of its siblings that come BEFORE it when the children are ordered by the projection of their box centres on the octant's
    // diagonal (ties by slot).  The traversal reads its octant's nibble and intersects it with the quad's hit mask (dbvh.h).
    {
        float cx[4], cy[4], cz[4];
        for (int k = 0; k < 4; ++k) { const BvhChild &ch = nd.child[k]; cx[k] = ch.x[0] + ch.x[1]; cy[k] = ch.y[0] + ch.y[1]; cz[k] = ch.z[0] + ch.z[1]; }
        for (int k = 0; k < ns; ++k) {
            uint32_t word = 0;
            for (int oct = 0; oct < 8; ++oct) {
                const float sx = (oct & 1) ? -1.f : 1.f, sy = (oct & 2) ? -1.f : 1.f, sz = (oct & 4) ? -1.f : 1.f;
                const float mine = sx * cx[k] + sy * cy[k] + sz * cz[k];
                uint32_t before = 0;
                for (int j = 0; j < ns; ++j) {
                    if (j == k) continue;
                    const float other = sx * cx[j] + sy * cy[j] + sz * cz[j];
                    if (other < mine || (other == mine && j < k)) before |= 1u << j;
                }
                word |= before << (4 * oct);
            }
            nd.child[k].pad = word;
        }
    }
    nodes[w.out] = nd;
    if (leaves) atomicAdd(n_leaves, leaves);
    atomicMax(max_need, w.acc + ns - 1);
}

// RGBFilm accumulation (frame_buffer.cpp:117-126): acc = lerp(1/(f+1), acc, L_f), frame by frame for every owned pixel
__global__ void k_film_resolve(RenderRest A) {
    uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= A.n_slots) return;
    uint32_t px, py;
    const uint32_t width = A.params->width;
    if (!slot_to_pixel(A, slot, width, A.params->height, &px, &py)) return;
    float4 a4 = A.accum[(size_t) py * width + px];
    V4 acc = {a4.x, a4.y, a4.z, a4.w};
    for (uint32_t f = 0; f < A.frame_count; ++f) {
        float4 l = A.stage[(size_t) f * A.n_slots + slot];
        float a = 1.f / (float) (A.frame_begin + f + 1u);
#ifdef VMK_DIAG
        V4 val = {l.x, l.y, l.z, l.w};
#else
        V4 val = {l.x, l.y, l.z, 1.f};
#endif
        acc = lerp4(a, acc, val);
    }
    A.accum[(size_t) py * width + px] = make_float4(acc.x, acc.y, acc.z, acc.w);
}

// ---------------------------------------------------------------------------------------------------------
// traversal replay, tone map, unit tests
// ---------------------------------------------------------------------------------------------------------
template<bool DEEP, bool ANYHIT>
__global__ __launch_bounds__(kBlock) void k_trace(const DScene S, uint32_t n, const float *org, const float *dir, const float *tmax,
                                                  int any_hit, uint32_t *hit_out, unsigned long long *counters, uint32_t *queue, uint32_t chunk) {
    __shared__ WaveScratch s_ws[kBlock / 64];
    WaveScratch *ws = s_ws + (threadIdx.x >> 6);
    DCounters cnt = {0, 0, 0, 0, 0, 0, 0};
    // persistent waves: the quads of every wave pull rays from one pool (SoA planes, 4 B/lane loads) until it is empty
    GlobalRayIO io = {chunk, org, dir, tmax, n, queue, reinterpret_cast<uint4 *>(hit_out), any_hit != 0, 0, 0, false};
    uint32_t n_rays = 0;
    if (S.n_tris == 0) { // nothing to hit
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { io.store((int) i, false, VMK_INVALID, VMK_INVALID, VMK_INVALID, 0.f, 0.f); ++n_rays; }
    } else traverse_core<GlobalRayIO, true, DEEP, ANYHIT>(S, io, ws, cnt, &n_rays);
    if (any_hit) cnt.shadow += n_rays; else cnt.closest += n_rays;
    uint32_t c[4] = {cnt.closest, cnt.shadow, cnt.nodes, cnt.tris};
    for (int k = 0; k < 4; ++k) { uint32_t s = wave_sum(c[k]); if ((threadIdx.x & 63) == 0 && s) atomicAdd(counters + k, (unsigned long long) s); }
}

__device__ __forceinline__ float tone1(uint32_t tm, float x) { // tonemapper/impl.cpp:16-45
    if (tm == 1) { float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f; return saturate_((x * (a * x + b)) / (x * (c * x + d) + e)); }
    if (tm == 2) return x / (x + 1.f);
    return x;
}
__global__ void k_tonemap(const float4 *accum, float4 *out, uint32_t n, float exposure, uint32_t tm, int final_picture) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = accum[i];
    float v[3] = {a.x, a.y, a.z};
    for (int c = 0; c < 3; ++c) {
        float x = 1.f - exp_(-v[c] * exposure); // frame_buffer.cpp:72-74
        x = tone1(tm, x);
        if (final_picture) { // Pipeline::final_picture applies the tone mapper again + sRGB (pipeline.cpp:337-354, postprocessor.cpp:13-30)
            x = tone1(tm, x);
            if (final_picture == 1) x = x <= 0.0031308f ? 12.92f * x : 1.055f * __powf(x, 1.f / 2.4f) - 0.055f; // (2: a name ending in exr / hdr — no gamma, pipeline.cpp:339)
        }
        v[c] = x;
    }
    out[i] = make_float4(v[0], v[1], v[2], 1.f);
}

__global__ void k_test(const DScene *scene, const vmk_render_params *P, uint32_t kind, uint32_t n, const float *in, uint32_t in_stride, float *out, uint32_t out_stride) {
    __shared__ WaveScratch s_ws[1]; // launched with 64-thread blocks
    __shared__ UnitState s_us[1];   // kind 6: the path state between vertices
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    if (!live && kind != 6 && kind != 7 && kind != 8) return; // kinds 6/7 trace rays: every lane of the wave has to stay
    const float *a = in + (size_t) (live ? i : 0) * in_stride;
    float *o = out + (size_t) (live ? i : 0) * out_stride;
    DCounters cnt = {0, 0, 0, 0, 0, 0, 0};
    switch (kind) {
        case 0: { Sampler s; s.start(f2u(a[0]), f2u(a[1]), f2u(a[2]), f2u(a[3])); for (int k = 0; k < 8; ++k) o[k] = s.next_1d(); break; }
        case 1: {
            float s, c; sincos_(a[0], &s, &c);
            o[0] = s; o[1] = c; o[2] = acos_(clamp_(a[0], -1.f, 1.f)); o[3] = atan2_(a[1], a[0]); o[4] = exp_(-abs_(a[0])); o[5] = sqrt_(abs_(a[0]));
            if (out_stride >= 7) o[6] = log_(abs_(a[0]) * 0.125f + 5.9604645e-8f);
            break;
        }
        case 2: {
            V2 u = {a[0], a[1]};
            V2 d = square_to_disk(u); V3 c = square_to_cosine_hemisphere(u); V2 t = square_to_triangle(u);
            o[0] = d.x; o[1] = d.y; o[2] = c.x; o[3] = c.y; o[4] = c.z; o[5] = t.x; o[6] = t.y; o[7] = sample_tent(a[0], 0.5f);
            break;
        }
        case 3: {
            V3 wo = normalize(mk3(a[0], a[1], a[2]));
            V3 wh = sample_wh(wo, {a[3], a[4]}, a[5], a[6]);
            o[0] = wh.x; o[1] = wh.y; o[2] = wh.z; o[3] = bsdf_D(wh, a[5], a[6]); o[4] = bsdf_G1(wo, a[5], a[6]);
            o[5] = PDF_wh(wo, wh, a[5], a[6]); o[6] = fresnel_dielectric(abs_dot(wo, wh), a[7]); o[7] = fresnel_complex(abs_dot(wo, wh), a[7], 3.5f);
            break;
        }
        case 4: {
            const DScene S = *scene;
            uint32_t mat_id = f2u(a[0]);
            Interaction it;
            it.pos = mk3(0.f); it.ng = mk3(0, 0, 1);
            it.shading = {mk3(1, 0, 0), mk3(0, 1, 0), mk3(0, 0, 1)};
            it.wo = normalize(mk3(a[4], a[5], a[6]));
            it.uv = {a[10], a[11]};
            it.mat_id = mat_id; it.light_id = VMK_INVALID; it.prim_id = 0; it.prim_area = 1.f;
            V3 wi = normalize(mk3(a[7], a[8], a[9]));
            it.shading = compute_shading_frame(S, S.materials + mat_id, it, cnt);
            MatCtx mc; mc.lobe_lds = lobe_lds_slot(s_ws); mat_prepare<true>(S, S.materials + mat_id, it, mc, cnt);
            Sampler smp; smp.start(f2u(a[1]), f2u(a[2]), f2u(a[3]), 1);
            ScatterEval se; BSDFSample bs;
            mat_evaluate_and_sample<true>(S, mc, it, wi, smp, se, bs, cnt);
            o[0] = se.f.x; o[1] = se.f.y; o[2] = se.f.z; o[3] = se.pdf; o[4] = u2f(se.flags);
            o[5] = bs.wi.x; o[6] = bs.wi.y; o[7] = bs.wi.z; o[8] = bs.eval.f.x; o[9] = bs.eval.f.y; o[10] = bs.eval.f.z; o[11] = bs.eval.pdf; o[12] = bs.eta;
            break;
        }
        case 5: {
            Sampler s; s.start(f2u(a[0]), f2u(a[1]), f2u(a[2]), 0);
            Ray r = generate_ray(P, f2u(a[0]), f2u(a[1]), s);
            o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
            break;
        }
        case 6: { // whole path of one (pixel, frame): 8 floats per vertex for up to 8 vertices, then L (3 floats)
            const DScene S = *scene;
            unit_path(S, P, s_ws, s_us, live, f2u(a[0]), f2u(a[1]), f2u(a[2]), o, cnt);
            break;
        }
#ifdef VMK_DIAG
        case 8: { // diagnostic build only: kind 6 through path_bounce<false, false> (single-lobe, no media)
            const DScene S = *scene;
            uint32_t px = f2u(a[0]), py = f2u(a[1]), frame = f2u(a[2]);
            Sampler smp; smp.start(px, py, frame, 0);
            PathState ps; ps.ray = generate_ray(P, px, py, smp);
            smp.start(px, py, frame, 1);
            path_begin(ps, P);
            bool alive = live;
            for (int v = 0; v < kUnitPathVertexCap && __any(alive); ++v) {
                float dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                int st = path_bounce<false, false>(S, P, s_ws, ps, smp, cnt, dbg, alive);
                if (alive && v < 8) for (int k = 0; k < 8; ++k) o[v * 8 + k] = dbg[k];
                if (st == kPathTail && alive) st = tail_is_primary(P, px, py, frame, ps.ray.d) ? kPathEnd : kPathGoOn;
                if (st != kPathGoOn) alive = false;
            }
            if (live) { o[64] = ps.L.x; o[65] = ps.L.y; o[66] = ps.L.z; }
            break;
        }
#endif
        case 7: { // ray capture of one (pixel, frame) for the traversal replay: o[0] = vertex count, then 16 floats per
                  // vertex [closest ray o.xyz d.xyz t_max, 1 | shadow ray o.xyz d.xyz t_max, traced]; needs out_stride >= 1 + 16 * 24
            const DScene S = *scene;
            uint32_t px = f2u(a[0]), py = f2u(a[1]), frame = f2u(a[2]);
            Sampler smp; smp.start(px, py, frame, 0);
            PathState ps; ps.ray = generate_ray(P, px, py, smp);
            smp.start(px, py, frame, 1);
            path_begin(ps, P);
            int nv = 0;
            bool alive = live && P->max_depth > 0;
            for (int v = 0; v < kUnitPathVertexCap && __any(alive); ++v) {
                float dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                Ray r = ps.ray;
                int st = path_bounce<true, true, true, true>(S, P, s_ws, ps, smp, cnt, dbg, alive);
                if (alive && v < 24) {
                    float *q = o + 1 + v * 16;
                    q[0] = r.o.x; q[1] = r.o.y; q[2] = r.o.z; q[3] = r.d.x; q[4] = r.d.y; q[5] = r.d.z; q[6] = r.t_max; q[7] = 1.f;
                    for (int k = 0; k < 8; ++k) q[8 + k] = dbg[8 + k];
                    nv = v + 1;
                }
                if (st == kPathTail && alive) st = tail_is_primary(P, px, py, frame, ps.ray.d) ? kPathEnd : kPathGoOn;
                if (st != kPathGoOn) alive = false;
            }
            if (live) o[0] = (float) nv;
            break;
        }
        default: break;
    }
}

// (the AOV pass — AovArgs, k_aov — lives in drender.h: one instance per spectrum, like the megakernel)

// ---------------------------------------------------------------------------------------------------------
// Albedo-table precompute (the reference's `vision-precompute` app, apps/precompute/main.cpp:24-41):
// Material::precompute_lobe (material.h:121-163) — texel (x, y, z) of a res^3 (res^2 for table 0) grid, ratio =
// idx / (res - 1), sampler.start((x, y), 0, 0), Lobe::precompute_with_radio + integral_albedo (lobe.cpp:13-33) in
// Importance mode, no energy compensation while measuring.  One lane integrates one texel with its own sequential
// RNG stream, exactly like the reference's kernel, accumulating in f64.
// which: 0 PureReflection (mirror.cpp:53-57, lobe.h:344-356), 1 Dielectric, 2 DielectricInv (glass.cpp:14-76),
//        3 Specular (principled_bsdf.cpp:177-190), 4 Coat (principled_bsdf.cpp:135-146)
// ---------------------------------------------------------------------------------------------------------
__global__ void k_albedo(uint32_t which, uint32_t res, uint32_t sample_num, float *out) {
    const uint32_t depth = which == 0 ? 1u : res, nc = (which == 1 || which == 2) ? 2u : 1u;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= res * res * depth) return;
    uint32_t x = i % res, y = (i / res) % res, z = i / (res * res);
    Sampler sampler; sampler.start(x, y, 0, 0);
    float rx = (float) x / (float) (res - 1), ry = (float) y / (float) (res - 1), rz = which == 0 ? 0.f : (float) z / (float) (res - 1);
    Lobe l;
    l.kind = LB_MICROFACET; l.kr = mk3(1.f); l.rs = mk3(0.f); l.A = 0.f; l.B = 0.f; l.compensate = false; l.weight = 1.f; l.sample_weight = 1.f;
    l.fr.kind = FR_CONSTANT; l.fr.a = mk3(0.f); l.fr.b = mk3(0.f); l.fr.eta = 1.f;
    float a = clamp_(sqr(rx), 0.001f, 1.f); // from_ratio_x lobe.cpp:183-185 / glass.cpp:36-40
    l.ax = l.ay = a;
    float cos_t = clamp_(ry, 1e-4f, 1.0f);   // from_ratio_y lobe.cpp:158-164
    V3 wo = mk3(sqrt_(1.f - sqr(cos_t)), 0.f, cos_t);
    switch (which) {
        case 0: break;
        case 1: l.kind = LB_DIELECTRIC; l.fr.kind = FR_DIELECTRIC; l.fr.eta = lerp_(rz, 1.003f, 5.f); break;
        case 2: l.kind = LB_DIELECTRIC; l.fr.kind = FR_DIELECTRIC; l.fr.eta = rcp(lerp_(rz, 1.003f, 5.f)); break;
        case 3: l.fr.kind = FR_SCHLICK; l.fr.a = mk3(0.04f); l.fr.eta = schlick_ior_from_F0(pow4(rz)); break;
        default: l.fr.kind = FR_DIELECTRIC; l.fr.eta = lerp_(rz, 1.003f, 4.f); break;
    }
    LobeLuts none{}; // the lobes measured here never touch the scene (compensate == false)
    double acc0 = 0.0, acc1 = 0.0;
    for (uint32_t k = 0; k < sample_num; ++k) {
        bool valid = true;
        V3 wi = sample_wi_local(l, wo, sampler, &valid);
        ScatterEval se; se.f = mk3(0.f); se.pdf = 0.f; se.flags = flag::Unset;
        if (l.kind == LB_DIELECTRIC) { // DielectricPrecompute::compensate() == false (glass.cpp:17), Importance mode
            bool refl = same_hemisphere(wo, wi);
            float eta = l.fr.eta, eta_p = refl ? 1.f : eta;
            V3 wh = face_forward(normalize(wo + wi * eta_p), wo);
            V3 F = l.fr.evaluate(abs_dot(wh, wo));
            if (refl) { se.f = F * BRDF_div_fr(wo, wh, wi, l.ax, l.ay); se.pdf = PDF_wi_reflection(wo, wh, l.ax, l.ay) * dielectric_refl_prob(l, F); }
            else {
                V3 wh2 = normalize(wo + wi * eta);
                se.f = ((1.f - F) * BTDF_div_ft(wo, wh2, wi, eta, l.ax, l.ay, false)) * l.kr;
                se.pdf = PDF_wi_transmission(wo, face_forward(wh, wo), wi, eta, l.ax, l.ay) * (1.f - dielectric_refl_prob(l, F));
            }
        } else {
            float eta_dummy;
            se = eval_local(none, l, wo, wi, &eta_dummy);
        }
        se.pdf *= valid ? 1.f : 0.f;
        if (se.pdf > 0.f) {
            float r = (se.f.x / se.pdf) * abs_cos_theta(wi);
            acc0 += (double) r;
            if (same_hemisphere(wi, wo)) acc1 += (double) r;
        }
    }
    out[(size_t) i * nc] = (float) (acc0 / sample_num);
    if (nc == 2) out[(size_t) i * nc + 1] = (float) (acc1 / sample_num);
}

// ---------------------------------------------------------------------------------------------------------
// context + C-ABI
// ---------------------------------------------------------------------------------------------------------
template<typename T>
struct DevBuf {
    T *p{nullptr};
    size_t n{0};
    hipError_t alloc(size_t count) { release(); n = count; if (!count) return hipSuccess; return hipMalloc((void **) &p, count * sizeof(T)); }
    hipError_t upload(const T *src, size_t count, hipStream_t s) {
        hipError_t e = alloc(count);
        if (e != hipSuccess || !count) return e;
        return hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s);
    }
    void release() { if (p) (void) hipFree(p); p = nullptr; n = 0; }
};

struct vmk_ctx {
    int device{0};
    hipStream_t stream{nullptr};
    hipEvent_t ev0{nullptr}, ev1{nullptr};
    std::string error;
    int n_cus{256};
    // scene
    bool scene_ready{false}, accel_ready{false}, params_ready{false};
    bool full_materials{true}; // scene has mix / principled_bsdf -> lobe-set variant of the megakernel
    int comm_world{0}, comm_rank{0}; // of the attached communicator (vmk_comm_init / vmk_comm_adopt)
    bool count_traversal{true}; // vmk_set_traversal_counters: launch the megakernel instance that tallies node fetches / triangle tests
    bool auto_self_check{true}, self_checked{false}, in_self_check{false}; // vmk_set_auto_self_check: the guard a host gets without asking for it
    bool hero{false};          // vmk_scene::spectrum == VMK_SPECTRUM_HERO -> the vmk_hero.hip instance of the megakernel
    bool hero4{false};         // ... with spectrum_dimension == 4 -> the vmk_hero4.hip instance
    DevBuf<float> rgb2spec, spd;
    uint32_t n_tris{0};
    DevBuf<vmk_tri_pos> tri_pos_in, tri_pos;
    DevBuf<vmk_tri_attr> tri_attr_in, tri_attr;
    DevBuf<uint32_t> tri_lookup;
    DevBuf<vmk_instance> instances;
    DevBuf<vmk_material> materials;
    DevBuf<vmk_light> lights;
    DevBuf<vmk_medium> mediums;
    uint32_t n_mediums{0};
    bool has_light_alias{false};
    DevBuf<vmk_texture> textures;
    DevBuf<uint8_t> tex_data;
    DevBuf<float> alias_prob, alias_func, srgb_lut, luts;
    DevBuf<uint32_t> alias_idx;
    DevBuf<BvhNode> nodes;
    DevBuf<uint32_t> stack_overflow; // only for trees deeper than the LDS stack
    DevBuf<DSceneFull> d_scene; // DSceneBase prefix + hero tail (dscene.h)
    DSceneFull h_scene{};
    float world_min[3]{}, world_max[3]{};
    vmk_accel_info accel{};
    // render state
    vmk_render_params params{};
    DevBuf<vmk_render_params> d_params;
    DevBuf<float4> own_fb, tm_out, stage;
    float4 *fb{nullptr};
    DevBuf<uint32_t> queue;
    DevBuf<unsigned long long> counters;
    // tile ownership table of the last sharded launch (include/vmk.h vmk_tiles)
    DevBuf<uint32_t> tile_table;
    uint32_t tt_key[5]{0, 0, 0, 0, 0}; // tiles_x, tiles_y, rank, world, count
    // multi-GPU (RCCL, loaded on first use)
    void *comm{nullptr};
    bool comm_owned{false};
    hipStream_t comm_stream{nullptr};
    hipEvent_t render_done{nullptr}, comm_done{nullptr};
    bool comm_pending{false};
    DevBuf<float4> gather_send, gather_recv;
    // asynchronous launch timing (vmk_collect_kernel_ms): one HIP-event pair per vmk_render_batch call since the last collect
    bool timing{false};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> time_pool;
    size_t time_used{0};
#ifdef VMK_DIAG
    DevBuf<float> diag;
#endif
};

static thread_local std::string g_null_error;

// ---------------------------------------------------------------------------------------------------------
// RCCL, loaded on first use: a single-GPU host never needs librccl.so to be present
// ---------------------------------------------------------------------------------------------------------
struct RcclApi {
    void *lib{nullptr};
    decltype(&ncclGetUniqueId) GetUniqueId{nullptr};
    decltype(&ncclCommInitRank) CommInitRank{nullptr};
    decltype(&ncclCommCount) CommCount{nullptr};
    decltype(&ncclCommUserRank) CommUserRank{nullptr};
    decltype(&ncclCommDestroy) CommDestroy{nullptr};
    decltype(&ncclAllReduce) AllReduce{nullptr};
    decltype(&ncclAllGather) AllGather{nullptr};
    decltype(&ncclGetErrorString) GetErrorString{nullptr};
};
static RcclApi g_rccl;
static bool rccl_load(std::string &err) {
    if (g_rccl.lib) return true;
    // a host that already runs RCCL (torch.distributed's "nccl" backend on ROCm) keeps one copy: reuse a loaded library first
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    void *h = nullptr;
    for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h) for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { err = std::string("RCCL not found (dlopen librccl.so): ") + (dlerror() ? dlerror() : "?"); return false; }
    RcclApi a; a.lib = h;
#define VMK_SYM(field, name) a.field = (decltype(a.field)) dlsym(h, name); if (!a.field) { err = std::string("RCCL symbol missing: ") + name; dlclose(h); return false; }
    VMK_SYM(GetUniqueId, "ncclGetUniqueId") VMK_SYM(CommInitRank, "ncclCommInitRank") VMK_SYM(CommCount, "ncclCommCount") VMK_SYM(CommUserRank, "ncclCommUserRank") VMK_SYM(CommDestroy, "ncclCommDestroy")
    VMK_SYM(AllReduce, "ncclAllReduce") VMK_SYM(AllGather, "ncclAllGather") VMK_SYM(GetErrorString, "ncclGetErrorString")
#undef VMK_SYM
    g_rccl = a;
    return true;
}
static void vmk_comm_release(vmk_ctx *ctx) {
    if (ctx->comm_stream) (void) hipStreamSynchronize(ctx->comm_stream);
    if (ctx->comm && ctx->comm_owned && g_rccl.CommDestroy) (void) g_rccl.CommDestroy((ncclComm_t) ctx->comm);
    ctx->comm = nullptr; ctx->comm_owned = false; ctx->comm_pending = false;
    if (ctx->render_done) (void) hipEventDestroy(ctx->render_done);
    if (ctx->comm_done) (void) hipEventDestroy(ctx->comm_done);
    if (ctx->comm_stream) (void) hipStreamDestroy(ctx->comm_stream);
    ctx->render_done = ctx->comm_done = nullptr; ctx->comm_stream = nullptr;
}
// gather / scatter of owned tiles for the all-gather form of the exchange: `table` lists, rank after rank, the tiles each rank
// owns (per_rank entries each, padded with VMK_INVALID); a tile is tile_size^2 float4 in row-major pixel order
__global__ void k_pack_tiles(const float4 *fb, float4 *send, const uint32_t *table, uint32_t n_owned, uint32_t tile_size, uint32_t tiles_x, uint32_t width, uint32_t height) {
    const uint32_t per_tile = tile_size * tile_size;
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t) n_owned * per_tile) return;
    uint32_t k = (uint32_t) (i / per_tile), r = (uint32_t) (i % per_tile), tile = table[k];
    uint32_t px = (tile % tiles_x) * tile_size + r % tile_size, py = (tile / tiles_x) * tile_size + r / tile_size;
    send[i] = (tile != VMK_INVALID && px < width && py < height) ? fb[(size_t) py * width + px] : make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void k_unpack_tiles(const float4 *recv, float4 *full, const uint32_t *table_all, uint32_t world, uint32_t per_rank, uint32_t tile_size, uint32_t tiles_x, uint32_t width, uint32_t height) {
    const uint32_t per_tile = tile_size * tile_size;
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t) world * per_rank * per_tile) return;
    uint32_t k = (uint32_t) (i / per_tile), r = (uint32_t) (i % per_tile), tile = table_all[k];
    if (tile == VMK_INVALID) return;
    uint32_t px = (tile % tiles_x) * tile_size + r % tile_size, py = (tile / tiles_x) * tile_size + r / tile_size;
    if (px < width && py < height) full[(size_t) py * width + px] = recv[i];
}

extern "C" {

uint32_t vmk_abi_version(void) { return VMK_ABI_VERSION; }
const char *vmk_last_error(const vmk_ctx *ctx) { return ctx ? ctx->error.c_str() : g_null_error.c_str(); }

int vmk_create(int device, vmk_ctx **out) {
    if (!out) return VMK_ERR_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { g_null_error = "vmk_create: no HIP device visible (this backend has no CPU fallback)"; return VMK_ERR_HIP; }
    if (device < 0 || device >= count) { g_null_error = "vmk_create: bad device index"; return VMK_ERR_ARG; }
    vmk_ctx *ctx = new vmk_ctx();
    ctx->device = device;
    auto bail = [&](hipError_t e, const char *what) { g_null_error = std::string(what) + ": " + hipGetErrorString(e); delete ctx; return VMK_ERR_HIP; };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cus = prop.multiProcessorCount;
    if ((e = ctx->queue.alloc(1)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = ctx->counters.alloc(8)) != hipSuccess) return bail(e, "hipMalloc");
    (void) hipMemsetAsync(ctx->counters.p, 0, 8 * sizeof(unsigned long long), ctx->stream);
    std::vector<float> srgb(256);
    for (int i = 0; i < 256; ++i) { double c = i / 255.0; srgb[i] = (float) (c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4)); }
    if ((e = ctx->srgb_lut.upload(srgb.data(), 256, ctx->stream)) != hipSuccess) return bail(e, "upload");
    (void) hipStreamSynchronize(ctx->stream);
    *out = ctx;
    return VMK_OK;
}

void vmk_destroy(vmk_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    ctx->tri_pos_in.release(); ctx->tri_pos.release(); ctx->tri_attr_in.release(); ctx->tri_attr.release(); ctx->tri_lookup.release();
    ctx->instances.release(); ctx->materials.release(); ctx->lights.release(); ctx->mediums.release(); ctx->textures.release(); ctx->tex_data.release();
    ctx->alias_prob.release(); ctx->alias_func.release(); ctx->alias_idx.release(); ctx->srgb_lut.release(); ctx->luts.release(); ctx->rgb2spec.release(); ctx->spd.release();
    ctx->nodes.release(); ctx->stack_overflow.release(); ctx->d_scene.release(); ctx->d_params.release(); ctx->own_fb.release(); ctx->tm_out.release(); ctx->stage.release();
    ctx->queue.release(); ctx->counters.release(); ctx->tile_table.release(); ctx->gather_send.release(); ctx->gather_recv.release();
    vmk_comm_release(ctx);
    for (auto &pr : ctx->time_pool) { (void) hipEventDestroy(pr.first); (void) hipEventDestroy(pr.second); }
    if (ctx->ev0) (void) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void) hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

void *vmk_stream(vmk_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

int vmk_upload_scene(vmk_ctx *ctx, const vmk_scene *sc) {
    if (!ctx) return VMK_ERR_ARG;
    if (!sc || sc->abi_version != VMK_ABI_VERSION) { ctx->error = "vmk_upload_scene: null scene or ABI version mismatch"; return VMK_ERR_ARG; }
    if (sc->n_tris == 0 || !sc->tri_pos || !sc->tri_attr || !sc->instances || !sc->materials || !sc->lights || sc->n_lights == 0) { ctx->error = "vmk_upload_scene: empty geometry / material / light tables"; return VMK_ERR_ARG; }
    if (sc->n_tris > kLeafFirstMask) { ctx->error = "vmk_upload_scene: too many triangles for the 28-bit leaf encoding"; return VMK_ERR_ARG; }
    // validate indices on the host: a kernel fault on this pool can reset the whole node
    for (uint32_t i = 0; i < sc->n_instances; ++i) {
        const vmk_instance &in = sc->instances[i];
        if ((in.mat_id != VMK_INVALID && in.mat_id >= sc->n_materials) || (in.light_id != VMK_INVALID && in.light_id >= sc->n_lights) || (uint64_t) in.tri_offset + in.tri_count > sc->n_tris) { ctx->error = "vmk_upload_scene: instance table references out-of-range rows"; return VMK_ERR_ARG; }
    }
    for (uint32_t i = 0; i < sc->n_tris; ++i) if (sc->tri_pos[i].inst >= sc->n_instances) { ctx->error = "vmk_upload_scene: triangle references a missing instance"; return VMK_ERR_ARG; }
    const bool hero = sc->spectrum == VMK_SPECTRUM_HERO;
    if (sc->spectrum != VMK_SPECTRUM_SRGB && !hero) { ctx->error = "vmk_upload_scene: unknown spectrum type"; return VMK_ERR_ARG; }
    auto spd_ok = [&](uint32_t off, uint32_t n) { return n >= 2 && (uint64_t) off + n <= sc->n_spd; };
    if (hero && sc->spectrum_dimension != 0 && sc->spectrum_dimension != 3 && sc->spectrum_dimension != 4) { ctx->error = "vmk_upload_scene: spectrum/hero is built for 3 or 4 wavelengths per path (vmk_scene::spectrum_dimension)"; return VMK_ERR_UNSUPPORTED; }
    if (hero) {
        if (!sc->rgb2spec || !sc->spd_data || sc->spd_cie_count < 2 || !(sc->spd_cie_interval > 0.f) || !(sc->cie_y_integral > 0.f)) { ctx->error = "vmk_upload_scene: hero spectrum tables missing"; return VMK_ERR_ARG; }
        for (int k = 0; k < 4; ++k) if (!spd_ok(sc->spd_cie[k], sc->spd_cie_count)) { ctx->error = "vmk_upload_scene: CIE tables out of range"; return VMK_ERR_ARG; }
        // SPD::eval reads samples [0, uint(470 / interval) + 1): must stay inside the table
        if ((uint32_t) (470.f / sc->spd_cie_interval) + 1u > sc->spd_cie_count) { ctx->error = "vmk_upload_scene: CIE table interval inconsistent"; return VMK_ERR_ARG; }
    }
    auto slot_ok = [&](const vmk_slot &s) { return s.tex == VMK_INVALID || (s.tex != VMK_SLOT_SPD && (s.tex & 0xffffu) < sc->n_textures); };
    auto spd_slot_ok = [&](const vmk_slot &s) { // "spd" nodes: metal eta / k and the ior of a dispersive glass, hero spectrum only
        uint32_t off, n; std::memcpy(&off, &s.v[0], 4); std::memcpy(&n, &s.v[1], 4);
        return hero && spd_ok(off, n) && s.v[2] > 0.f && (uint32_t) (470.f / s.v[2]) + 1u <= n;
    };
    for (uint32_t i = 0; i < sc->n_materials; ++i) {
        const vmk_material &m = sc->materials[i];
        if (m.type > VMK_MAT_PLASTIC) { ctx->error = "vmk_upload_scene: unknown material type"; return VMK_ERR_ARG; }
        if ((m.flags & VMK_MATF_HAS_NORMAL) && (m.normal.tex == VMK_SLOT_SPD || !slot_ok(m.normal) || (m.normal.tex != VMK_INVALID && (m.normal.tex & VMK_SLOT_TINTED)) || m.type == VMK_MAT_MIX || m.type == VMK_MAT_ADD)) { ctx->error = "vmk_upload_scene: bad normal slot"; return VMK_ERR_ARG; }
        for (int k = 0; k < VMK_MAX_SLOTS; ++k) {
            const vmk_slot &s = m.slot[k];
            const bool spd_allowed = (m.type == VMK_MAT_METAL && k < 2) || (m.type == VMK_MAT_GLASS && k == 1);
            if (s.tex == VMK_SLOT_SPD ? !(spd_allowed && spd_slot_ok(s)) : !slot_ok(s)) { ctx->error = "vmk_upload_scene: material slot references a missing texture or spectrum"; return VMK_ERR_ARG; }
        }
        if ((m.flags & VMK_MATF_DISPERSIVE) && !(m.type == VMK_MAT_GLASS && m.slot[1].tex == VMK_SLOT_SPD)) { ctx->error = "vmk_upload_scene: VMK_MATF_DISPERSIVE without a tabulated ior"; return VMK_ERR_ARG; }
        if (m.type == VMK_MAT_MIX || m.type == VMK_MAT_ADD) {
            if (m.child0 >= sc->n_materials || m.child1 >= sc->n_materials) { ctx->error = "vmk_upload_scene: mix child out of range"; return VMK_ERR_ARG; }
            uint32_t t0 = sc->materials[m.child0].type, t1 = sc->materials[m.child1].type;
            const bool ok0 = VMK_MAT_IS_SINGLE_LOBE(t0) || t0 == VMK_MAT_PRINCIPLED, ok1 = VMK_MAT_IS_SINGLE_LOBE(t1) || t1 == VMK_MAT_PRINCIPLED;
            if (!ok0 || !ok1 || (t0 == VMK_MAT_PRINCIPLED && t1 == VMK_MAT_PRINCIPLED)) { ctx->error = "vmk_upload_scene: mix / add children must be single-lobe materials, or one principled_bsdf next to a single-lobe material"; return VMK_ERR_UNSUPPORTED; }
            if ((sc->materials[m.child0].flags | sc->materials[m.child1].flags) & VMK_MATF_HAS_NORMAL) { ctx->error = "vmk_upload_scene: normal maps on the children of mix / add are not supported (per-lobe shading frames)"; return VMK_ERR_UNSUPPORTED; }
        }
        if (m.type == VMK_MAT_PRINCIPLED && !sc->luts.sheen_approx && (m.slot[VMK_P_SHEEN_WEIGHT].tex != VMK_INVALID || m.slot[VMK_P_SHEEN_WEIGHT].v[0] != 0.f)) { ctx->error = "vmk_upload_scene: sheen needs the LTC tables"; return VMK_ERR_UNSUPPORTED; }
    }
    for (uint32_t i = 0; i < sc->n_textures; ++i) {
        const vmk_texture &t = sc->textures[i];
        uint64_t bytes = (uint64_t) t.width * t.height * (t.format == VMK_TEX_RGBA32F ? 16u : 4u);
        if (t.width == 0 || t.height == 0 || t.format > VMK_TEX_RGBA32F || t.offset % 16 || t.offset + bytes > sc->tex_bytes) { ctx->error = "vmk_upload_scene: texture descriptor out of range"; return VMK_ERR_ARG; }
    }
    for (uint32_t i = 0; i < sc->n_lights; ++i) {
        const vmk_light &l = sc->lights[i];
        if (!slot_ok(l.color)) { ctx->error = "vmk_upload_scene: light colour references a missing texture"; return VMK_ERR_ARG; }
        if (l.type == VMK_LIGHT_AREA) {
            if (l.inst_id >= sc->n_instances || (uint64_t) l.alias_offset + l.alias_count > sc->n_alias || l.alias_count != sc->instances[l.inst_id].tri_count || l.alias_count == 0) { ctx->error = "vmk_upload_scene: area light tables inconsistent"; return VMK_ERR_ARG; }
        } else if (l.type == VMK_LIGHT_SPHERICAL) {
            if (l.res_x == 0 || l.res_y == 0 || l.alias_count != l.res_y || (uint64_t) l.alias_offset + l.alias_count > sc->n_alias || (uint64_t) l.cond_offset + (uint64_t) l.res_x * l.res_y > sc->n_alias) { ctx->error = "vmk_upload_scene: environment light tables inconsistent"; return VMK_ERR_ARG; }
        } else if (l.type == VMK_LIGHT_POINT || l.type == VMK_LIGHT_SPOT) {
            if (l.type == VMK_LIGHT_SPOT && !(l.cos_falloff_start > l.cos_angle)) { ctx->error = "vmk_upload_scene: spot light cone is empty"; return VMK_ERR_ARG; }
        } else if (l.type == VMK_LIGHT_PROJECTOR) {
            if (!(l.tan_xy[0] > 0.f) || !(l.tan_xy[1] > 0.f) || (l.color.tex != VMK_INVALID && (l.color.tex & VMK_SLOT_TINTED))) { ctx->error = "vmk_upload_scene: projector light frustum / image slot invalid"; return VMK_ERR_ARG; }
        } else { ctx->error = "vmk_upload_scene: unknown light type"; return VMK_ERR_ARG; }
    }
    if (sc->light_alias_offset != VMK_INVALID && (uint64_t) sc->light_alias_offset + sc->n_lights > sc->n_alias) { ctx->error = "vmk_upload_scene: light-sampler alias table out of range"; return VMK_ERR_ARG; }
    if (sc->n_mediums && !sc->mediums) { ctx->error = "vmk_upload_scene: mediums missing"; return VMK_ERR_ARG; }
    for (uint32_t i = 0; i < sc->n_instances; ++i) {
        const vmk_instance &in = sc->instances[i];
        if ((in.inside_medium != VMK_INVALID && in.inside_medium >= sc->n_mediums) || (in.outside_medium != VMK_INVALID && in.outside_medium >= sc->n_mediums)) { ctx->error = "vmk_upload_scene: instance medium id out of range"; return VMK_ERR_ARG; }
    }
    for (uint32_t i = 0; i < sc->n_alias; ++i) if (sc->alias_idx[i] >= sc->n_alias) { ctx->error = "vmk_upload_scene: alias index out of range"; return VMK_ERR_ARG; }
    if (sc->env_light != VMK_INVALID && (sc->env_light >= sc->n_lights || sc->lights[sc->env_light].type != VMK_LIGHT_SPHERICAL)) { ctx->error = "vmk_upload_scene: env_light is not a spherical light"; return VMK_ERR_ARG; }
    if (!sc->luts.pure_reflection || !sc->luts.dielectric || !sc->luts.dielectric_inv || !sc->luts.specular || !sc->luts.coat) { ctx->error = "vmk_upload_scene: albedo tables missing"; return VMK_ERR_ARG; }

    ctx->full_materials = false;
    for (uint32_t i = 0; i < sc->n_materials; ++i) if (!VMK_MAT_IS_SINGLE_LOBE(sc->materials[i].type) || (sc->materials[i].flags & VMK_MATF_HAS_NORMAL)) ctx->full_materials = true;
    if (const char *v = getenv("VMK_FORCE_FULL")) if (v[0] == '1') ctx->full_materials = true;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ctx->n_tris = sc->n_tris;
    HIP_TRY(ctx->tri_pos_in.upload(sc->tri_pos, sc->n_tris, st));
    HIP_TRY(ctx->tri_attr_in.upload(sc->tri_attr, sc->n_tris, st));
    HIP_TRY(ctx->instances.upload(sc->instances, sc->n_instances, st));
    HIP_TRY(ctx->materials.upload(sc->materials, sc->n_materials, st));
    HIP_TRY(ctx->lights.upload(sc->lights, sc->n_lights, st));
    HIP_TRY(ctx->mediums.upload(sc->mediums, sc->n_mediums, st));
    ctx->n_mediums = sc->n_mediums;
    HIP_TRY(ctx->textures.upload(sc->textures, sc->n_textures, st));
    HIP_TRY(ctx->tex_data.upload(sc->tex_data, (size_t) sc->tex_bytes, st));
    HIP_TRY(ctx->alias_prob.upload(sc->alias_prob, sc->n_alias, st));
    HIP_TRY(ctx->alias_idx.upload(sc->alias_idx, sc->n_alias, st));
    HIP_TRY(ctx->alias_func.upload(sc->alias_func, sc->n_alias, st));
    const uint32_t N = VMK_LUT_RES;
    const size_t sizes[6] = {(size_t) N * N, (size_t) N * N * N * 2, (size_t) N * N * N * 2, (size_t) N * N * N, (size_t) N * N * N, (size_t) N * N * 4};
    const float *src[6] = {sc->luts.pure_reflection, sc->luts.dielectric, sc->luts.dielectric_inv, sc->luts.specular, sc->luts.coat, sc->luts.sheen_approx};
    size_t total = 0; for (int i = 0; i < 6; ++i) total += sizes[i];
    HIP_TRY(ctx->luts.alloc(total));
    size_t off[6], o = 0;
    for (int i = 0; i < 6; ++i) { off[i] = o; if (src[i]) HIP_TRY(hipMemcpyAsync(ctx->luts.p + o, src[i], sizes[i] * 4, hipMemcpyHostToDevice, st)); o += sizes[i]; }
    ctx->hero = hero; ctx->hero4 = hero && sc->spectrum_dimension == 4;
    if (hero) {
        HIP_TRY(ctx->rgb2spec.upload(sc->rgb2spec, (size_t) 3 * VMK_RGB2SPEC_RES * VMK_RGB2SPEC_RES * VMK_RGB2SPEC_RES * 4, st));
        HIP_TRY(ctx->spd.upload(sc->spd_data, sc->n_spd, st));
    }
    DSceneFull &h = ctx->h_scene;
    h = DSceneFull{};
    if (hero) {
        h.hero.rgb2spec = ctx->rgb2spec.p; h.hero.spd_data = ctx->spd.p;
        for (int k = 0; k < 4; ++k) h.hero.spd_cie[k] = sc->spd_cie[k];
        h.hero.spd_cie_interval = sc->spd_cie_interval; h.hero.cie_y_integral = sc->cie_y_integral;
    }
    h.instances = ctx->instances.p; h.materials = ctx->materials.p; h.lights = ctx->lights.p; h.mediums = ctx->mediums.p; h.textures = ctx->textures.p; h.tex_data = ctx->tex_data.p;
    h.alias_prob = ctx->alias_prob.p; h.alias_idx = ctx->alias_idx.p; h.alias_func = ctx->alias_func.p; h.srgb_lut = ctx->srgb_lut.p;
    h.lut_pure_reflection = ctx->luts.p + off[0]; h.lut_dielectric = ctx->luts.p + off[1]; h.lut_dielectric_inv = ctx->luts.p + off[2];
    h.lut_specular = ctx->luts.p + off[3]; h.lut_coat = ctx->luts.p + off[4]; h.lut_sheen_approx = src[5] ? ctx->luts.p + off[5] : nullptr;
    h.n_tris = sc->n_tris; h.n_lights = sc->n_lights; h.env_light = sc->env_light;
    h.light_alias_offset = sc->light_alias_offset; h.light_alias_integral = sc->light_alias_integral;
    ctx->has_light_alias = sc->light_alias_offset != VMK_INVALID;
    for (int k = 0; k < 3; ++k) { ctx->world_min[k] = sc->world_min[k]; ctx->world_max[k] = sc->world_max[k]; }
    HIP_TRY(hipStreamSynchronize(st));
    ctx->scene_ready = true; ctx->accel_ready = false;
    return VMK_OK;
}

int vmk_build_accel(vmk_ctx *ctx) {
    if (!ctx) return VMK_ERR_ARG;
    if (!ctx->scene_ready) { ctx->error = "vmk_build_accel: no scene uploaded"; return VMK_ERR_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint32_t n = ctx->n_tris;
    const int nb = (int) ((n + 255) / 256);
    DevBuf<uint64_t> keys, keys_sorted;
    DevBuf<uint32_t> vals, vals_sorted, n_leaves;
    DevBuf<int> scalars; // [0] worst-case stack need, [1] BVH4 node counter, [2] next-level queue length
    DevBuf<uint8_t> temp, scan_temp;
    DevBuf<Box6> box; DevBuf<int> cl_a, cl_b, nn, pl_left, pl_right, pl_parent, pl_count;
    DevBuf<unsigned long long> flags, scan;
    DevBuf<vmk_tri_pos> pos_final; DevBuf<vmk_tri_attr> attr_final;
    DevBuf<Bvh4Work> q_a, q_b;
    auto cleanup = [&]() {
        keys.release(); keys_sorted.release(); vals.release(); vals_sorted.release(); n_leaves.release(); scalars.release(); temp.release(); scan_temp.release();
        box.release(); cl_a.release(); cl_b.release(); nn.release(); pl_left.release(); pl_right.release(); pl_parent.release(); pl_count.release();
        flags.release(); scan.release(); pos_final.release(); attr_final.release(); q_a.release(); q_b.release();
    };
#define BUILD_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ctx->error = std::string(#expr) + ": " + hipGetErrorString(e_); cleanup(); return VMK_ERR_HIP; } } while (0)
    BUILD_TRY(keys.alloc(n)); BUILD_TRY(keys_sorted.alloc(n)); BUILD_TRY(vals.alloc(n)); BUILD_TRY(vals_sorted.alloc(n));
    BUILD_TRY(ctx->tri_pos.alloc(n)); BUILD_TRY(ctx->tri_attr.alloc(n)); BUILD_TRY(ctx->tri_lookup.alloc(n));
    const size_t n_int = n > 1 ? n - 1 : 1; // a BVH4 has at most as many nodes as the binary tree has internal nodes
    BUILD_TRY(scalars.alloc(3)); BUILD_TRY(n_leaves.alloc(1));
    BUILD_TRY(ctx->nodes.alloc(n_int));
    BUILD_TRY(hipMemsetAsync(scalars.p, 0, 3 * sizeof(int), st));
    BUILD_TRY(hipMemsetAsync(n_leaves.p, 0, sizeof(uint32_t), st));
    BUILD_TRY(hipEventRecord(ctx->ev0, st));
    // ---- 1. Morton order ----
    float ext[3];
    for (int k = 0; k < 3; ++k) { ext[k] = ctx->world_max[k] - ctx->world_min[k]; ext[k] = ext[k] > 0.f ? 1.f / ext[k] : 0.f; }
    hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, st, ctx->tri_pos_in.p, n, make_float3(ctx->world_min[0], ctx->world_min[1], ctx->world_min[2]), make_float3(ext[0], ext[1], ext[2]), keys.p, vals.p);
    size_t temp_bytes = 0;
    BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys.p, keys_sorted.p, vals.p, vals_sorted.p, (int) n, 0, 63, st));
    BUILD_TRY(temp.alloc(temp_bytes ? temp_bytes : 16));
    BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, keys.p, keys_sorted.p, vals.p, vals_sorted.p, (int) n, 0, 63, st));
    hipLaunchKernelGGL(k_reorder, dim3(nb), dim3(256), 0, st, vals_sorted.p, n, ctx->tri_pos_in.p, ctx->tri_attr_in.p, ctx->tri_pos.p, ctx->tri_attr.p, ctx->tri_lookup.p);
    int h_scalars[3] = {0, 0, 0};
    uint32_t h_leaves = 1, h_nodes = 0;
    if (n > (uint32_t) kMaxLeafTris) {
        // ---- 2. PLOC binary hierarchy over the Morton-sorted triangles (ctx->tri_pos / tri_attr hold the sorted order here) ----
        int radius = 128; // classroom: 14.6 node visits per ray at radius 16, 13.1 at 128 (+6 % Mrays/s) for 4 ms more build time
        if (const char *r = getenv("VMK_PLOC_RADIUS")) radius = std::max(1, std::min(1024, atoi(r)));
        size_t n_all = 2 * (size_t) n;
        BUILD_TRY(box.alloc(n_all)); BUILD_TRY(cl_a.alloc(n)); BUILD_TRY(cl_b.alloc(n)); BUILD_TRY(nn.alloc(n));
        BUILD_TRY(pl_left.alloc(n_all)); BUILD_TRY(pl_right.alloc(n_all)); BUILD_TRY(pl_parent.alloc(n_all)); BUILD_TRY(pl_count.alloc(n_all));
        BUILD_TRY(flags.alloc(n)); BUILD_TRY(scan.alloc(n)); BUILD_TRY(pos_final.alloc(n)); BUILD_TRY(attr_final.alloc(n));
        BUILD_TRY(q_a.alloc(n_int)); BUILD_TRY(q_b.alloc(n_int));
        size_t scan_bytes = 0;
        BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flags.p, scan.p, (int) n, st));
        BUILD_TRY(scan_temp.alloc(scan_bytes ? scan_bytes : 16));
        hipLaunchKernelGGL(k_ploc_init, dim3(nb), dim3(256), 0, st, ctx->tri_pos.p, (int) n, box.p, cl_a.p, pl_parent.p, pl_count.p);
        int m = (int) n, node_base = (int) n, rounds = 0;
        int *cl_in = cl_a.p, *cl_out = cl_b.p;
        while (m > 1) {
            int mb = (m + 255) / 256;
            int force = 0;
            for (;;) {
                if (!force) hipLaunchKernelGGL(k_ploc_nn, dim3(mb), dim3(256), 0, st, m, cl_in, box.p, radius, nn.p);
                hipLaunchKernelGGL(k_ploc_flags, dim3(mb), dim3(256), 0, st, m, nn.p, flags.p, force);
                BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(scan_temp.p, scan_bytes, flags.p, scan.p, m, st));
                unsigned long long last_f = 0, last_s = 0;
                BUILD_TRY(hipMemcpyAsync(&last_f, flags.p + (m - 1), 8, hipMemcpyDeviceToHost, st));
                BUILD_TRY(hipMemcpyAsync(&last_s, scan.p + (m - 1), 8, hipMemcpyDeviceToHost, st));
                BUILD_TRY(hipStreamSynchronize(st));
                unsigned long long tot = last_f + last_s;
                int merges = (int) (tot >> 32), m_next = (int) (tot & 0xffffffffull);
                if (merges == 0 && !force) { force = 1; continue; } // tie pathologies: pair neighbours (i, i^1) so every round makes progress
                hipLaunchKernelGGL(k_ploc_merge, dim3(mb), dim3(256), 0, st, m, cl_in, nn.p, flags.p, scan.p, node_base, box.p, pl_left.p, pl_right.p, pl_parent.p, pl_count.p, cl_out, force);
                node_base += merges; m = m_next;
                break;
            }
            std::swap(cl_in, cl_out);
            if (++rounds > 4096) { ctx->error = "vmk_build_accel: PLOC did not converge"; cleanup(); return VMK_ERR_STATE; }
        }
        if (node_base != (int) (2 * n - 1)) { ctx->error = "vmk_build_accel: PLOC node count mismatch"; cleanup(); return VMK_ERR_STATE; }
        // ---- 3. depth-first triangle order ----
        hipLaunchKernelGGL(k_ploc_place, dim3(nb), dim3(256), 0, st, (int) n, pl_left.p, pl_right.p, pl_parent.p, pl_count.p, vals_sorted.p,
                           ctx->tri_pos.p, ctx->tri_attr.p, pos_final.p, attr_final.p, ctx->tri_lookup.p);
        BUILD_TRY(hipMemcpyAsync(ctx->tri_pos.p, pos_final.p, (size_t) n * sizeof(vmk_tri_pos), hipMemcpyDeviceToDevice, st));
        BUILD_TRY(hipMemcpyAsync(ctx->tri_attr.p, attr_final.p, (size_t) n * sizeof(vmk_tri_attr), hipMemcpyDeviceToDevice, st));
        // ---- 4. collapse to 128 B BVH4 nodes, level by level from the root (binary id 2n-2 -> node 0) ----
        Bvh4Work root_work = {(int) (2 * n - 2), 0, 0, 0};
        int one = 1;
        BUILD_TRY(hipMemcpyAsync(q_a.p, &root_work, sizeof(root_work), hipMemcpyHostToDevice, st));
        BUILD_TRY(hipMemcpyAsync(scalars.p + 1, &one, sizeof(int), hipMemcpyHostToDevice, st));
        Bvh4Work *q_in = q_a.p, *q_out = q_b.p;
        int n_in = 1, levels = 0;
        while (n_in > 0) {
            BUILD_TRY(hipMemsetAsync(scalars.p + 2, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_bvh4_level, dim3((n_in + 255) / 256), dim3(256), 0, st, q_in, n_in, q_out, scalars.p + 2, scalars.p + 1, box.p,
                               pl_left.p, pl_right.p, pl_count.p, ctx->nodes.p, n_leaves.p, scalars.p);
            BUILD_TRY(hipMemcpyAsync(&n_in, scalars.p + 2, sizeof(int), hipMemcpyDeviceToHost, st));
            BUILD_TRY(hipStreamSynchronize(st));
            std::swap(q_in, q_out);
            if (++levels > 4096) { ctx->error = "vmk_build_accel: BVH4 collapse did not terminate"; cleanup(); return VMK_ERR_STATE; }
        }
        BUILD_TRY(hipMemcpyAsync(h_scalars, scalars.p, sizeof(h_scalars), hipMemcpyDeviceToHost, st));
        BUILD_TRY(hipMemcpyAsync(&h_leaves, n_leaves.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        BUILD_TRY(hipStreamSynchronize(st));
        h_nodes = (uint32_t) h_scalars[1];
    }
    BUILD_TRY(hipEventRecord(ctx->ev1, st));
    BUILD_TRY(hipStreamSynchronize(st));
    float ms = 0.f; (void) hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    cleanup();
#undef BUILD_TRY
    if (h_nodes > n_int) { ctx->error = "vmk_build_accel: BVH4 node count exceeds the allocation"; return VMK_ERR_STATE; }
    if ((uint64_t) n * sizeof(vmk_tri_pos) > 0xffffffffull) { ctx->error = "vmk_build_accel: triangle array exceeds the 4 GiB the traversal's 32-bit offsets address"; return VMK_ERR_UNSUPPORTED; }
    if ((uint64_t) h_nodes * sizeof(BvhNode) > 0xffffffffull) { ctx->error = "vmk_build_accel: node array exceeds the 4 GiB the traversal's 32-bit node offsets address"; return VMK_ERR_UNSUPPORTED; }
    if (h_scalars[0] > kQuadStack + kStackOverflow) { ctx->error = "vmk_build_accel: worst-case traversal stack need " + std::to_string(h_scalars[0]) + " exceeds the per-ray stack (" + std::to_string(kQuadStack) + " entries in LDS + " + std::to_string(kStackOverflow) + " in HBM)"; return VMK_ERR_UNSUPPORTED; }
    ctx->stack_overflow.release();
    if (h_scalars[0] > kQuadStack) HIP_TRY(ctx->stack_overflow.alloc((size_t) kOverflowWaves * kStackOverflow * 64)); // deep tree: HBM overflow of the LDS stacks (dbvh.h)
    DSceneFull &h = ctx->h_scene;
    h.tri_pos = ctx->tri_pos.p; h.tri_attr = ctx->tri_attr.p; h.tri_lookup = ctx->tri_lookup.p; h.nodes = ctx->nodes.p;
    h.stack_overflow = ctx->stack_overflow.p;
    h.root = n <= (uint32_t) kMaxLeafTris ? (int32_t) ~((uint32_t) 0 | ((n - 1u) << 28)) : 0;
    HIP_TRY(ctx->d_scene.upload(&h, 1, st));
    HIP_TRY(hipStreamSynchronize(st));
    ctx->tri_pos_in.release(); ctx->tri_attr_in.release();
    ctx->scene_ready = false; // host copies consumed; a new upload is needed before rebuilding
    ctx->accel = {h_nodes, h_leaves, (uint32_t) sizeof(BvhNode), (uint32_t) sizeof(vmk_tri_pos), ms, (uint32_t) h_scalars[0], (uint32_t) kQuadStack};
    ctx->accel_ready = true;
    ctx->self_checked = false;
    return VMK_OK;
}

int vmk_accel_info_get(vmk_ctx *ctx, vmk_accel_info *out) {
    if (!ctx || !out) return VMK_ERR_ARG;
    if (!ctx->accel_ready) { ctx->error = "vmk_accel_info_get: accel not built"; return VMK_ERR_STATE; }
    *out = ctx->accel;
    return VMK_OK;
}

int vmk_set_render_params(vmk_ctx *ctx, const vmk_render_params *p) {
    if (!ctx) return VMK_ERR_ARG;
    if (!p || p->width == 0 || p->height == 0 || (uint64_t) p->width * p->height > (1ull << 30)) { ctx->error = "vmk_set_render_params: bad resolution"; return VMK_ERR_ARG; }
    if (p->filter_type > VMK_FILTER_TABLE || p->mis_mode > 2 || p->light_sampler > 1) { ctx->error = "vmk_set_render_params: bad filter / mis mode / light sampler"; return VMK_ERR_ARG; }
    if (p->filter_type == VMK_FILTER_TABLE) {
        for (uint32_t i = 0; i < VMK_FILTER_TABLE_SIZE; ++i) if (p->filter_marginal_alias[i] >= VMK_FILTER_TABLE_SIZE) { ctx->error = "vmk_set_render_params: filter alias out of range"; return VMK_ERR_ARG; }
        for (uint32_t i = 0; i < VMK_FILTER_TABLE_SIZE * VMK_FILTER_TABLE_SIZE; ++i) if (p->filter_cond_alias[i] >= VMK_FILTER_TABLE_SIZE) { ctx->error = "vmk_set_render_params: filter alias out of range"; return VMK_ERR_ARG; }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    bool res_changed = !ctx->params_ready || ctx->params.width != p->width || ctx->params.height != p->height;
    ctx->params = *p;
    HIP_TRY(ctx->d_params.upload(&ctx->params, 1, ctx->stream));
    if (res_changed) {
        size_t n = (size_t) p->width * p->height;
        if (ctx->fb == ctx->own_fb.p) ctx->fb = nullptr;
        HIP_TRY(ctx->own_fb.alloc(n));
        HIP_TRY(ctx->tm_out.alloc(n));
        HIP_TRY(hipMemsetAsync(ctx->own_fb.p, 0, n * sizeof(float4), ctx->stream));
        ctx->fb = ctx->own_fb.p;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->params_ready = true;
    ctx->self_checked = false; // (the parameters select the kernel variant)
    return VMK_OK;
}

int vmk_set_framebuffer(vmk_ctx *ctx, void *fb_device) {
    if (!ctx) return VMK_ERR_ARG;
    if (!ctx->params_ready) { ctx->error = "vmk_set_framebuffer: set render params first"; return VMK_ERR_STATE; }
    ctx->fb = fb_device ? (float4 *) fb_device : ctx->own_fb.p;
    return VMK_OK;
}

int vmk_reset_accum(vmk_ctx *ctx) {
    if (!ctx) return VMK_ERR_ARG;
    if (!ctx->params_ready || !ctx->fb) { ctx->error = "vmk_reset_accum: no framebuffer"; return VMK_ERR_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->comm_pending) { HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->comm_done, 0)); ctx->comm_pending = false; }
    HIP_TRY(hipMemsetAsync(ctx->fb, 0, (size_t) ctx->params.width * ctx->params.height * sizeof(float4), ctx->stream));
    return VMK_OK;
}

int vmk_render_batch(vmk_ctx *ctx, uint32_t frame_begin, uint32_t frame_count, const vmk_tiles *tiles, float *kernel_ms) {
    if (!ctx) return VMK_ERR_ARG;
    if (!ctx->accel_ready || !ctx->params_ready || !ctx->fb) { ctx->error = "vmk_render_batch: scene/accel/params not ready"; return VMK_ERR_STATE; }
    if (ctx->params.process_mediums && ctx->params.camera_medium != VMK_INVALID && ctx->params.camera_medium >= ctx->n_mediums) { ctx->error = "vmk_render_batch: camera medium out of range"; return VMK_ERR_ARG; }
    if (ctx->params.light_sampler == 1 && !ctx->has_light_alias) { ctx->error = "vmk_render_batch: the power light sampler needs vmk_scene::light_alias_offset"; return VMK_ERR_ARG; }
    if (kernel_ms) *kernel_ms = 0.f;
    if (frame_count == 0) return VMK_OK;
    if (ctx->auto_self_check && !ctx->self_checked && !ctx->in_self_check) {
        // the first batch after a build / a parameter change: the megakernel variant this scene selects against its separately compiled twin
        // on 256 pixels of frame 0 (vmk_self_check) — the guard against the toolchain's miscompiles that a Vision-side user gets by default
        ctx->in_self_check = true;
        uint32_t n = 0, bad = 0;
        const int rc = vmk_self_check(ctx, 256, &n, &bad);
        ctx->in_self_check = false;
        if (rc != VMK_OK) return rc;
        ctx->self_checked = true;
    }
    RenderArgs A{};
    A.scene = ctx->h_scene; A.params = ctx->d_params.p; A.accum = ctx->fb; A.queue = ctx->queue.p; A.counters = ctx->counters.p;
    uint32_t ts = kDefaultTile, rank = 0, world = 1;
    if (tiles && tiles->tile_size) {
        ts = tiles->tile_size; rank = tiles->rank; world = tiles->world ? tiles->world : 1;
        if ((ts & (ts - 1)) || ts > 1024 || rank >= world) { ctx->error = "vmk_render_batch: tile_size must be a power of two <= 1024 and rank < world"; return VMK_ERR_ARG; }
    }
    uint32_t shift = 0; while ((1u << shift) < ts) ++shift;
    A.tile_size = ts; A.tile_shift = shift;
    A.tiles_x = (ctx->params.width + ts - 1) / ts; A.tiles_y = (ctx->params.height + ts - 1) / ts;
    uint32_t n_tiles = A.tiles_x * A.tiles_y;
    uint32_t owned = n_tiles;
    A.rank = rank; A.world = world; A.tile_table = nullptr;
    if (world > 1) { // owner(tx, ty) = (tx + skew * ty) mod world: the owned tiles, ascending, as a device table
        if (ctx->tt_key[0] != A.tiles_x || ctx->tt_key[1] != A.tiles_y || ctx->tt_key[2] != rank || ctx->tt_key[3] != world) {
            std::vector<uint32_t> table = owned_tiles(A.tiles_x, A.tiles_y, rank, world);
            HIP_TRY(hipStreamSynchronize(ctx->stream)); // a launch in flight may still read the old table
            HIP_TRY(ctx->tile_table.upload(table.data(), table.size(), ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            ctx->tt_key[0] = A.tiles_x; ctx->tt_key[1] = A.tiles_y; ctx->tt_key[2] = rank; ctx->tt_key[3] = world; ctx->tt_key[4] = (uint32_t) table.size();
        }
        owned = ctx->tt_key[4];
        A.tile_table = ctx->tile_table.p;
    }
    uint64_t n_slots = (uint64_t) owned * ts * ts;
    if (n_slots == 0) return VMK_OK;
    if (n_slots > 0x7fffffffull) { ctx->error = "vmk_render_batch: too many pixels"; return VMK_ERR_ARG; }
    A.n_slots = (uint32_t) n_slots;
    HIP_TRY(hipSetDevice(ctx->device));
    // frames per launch: bounded by the staging planes (<= 8 GiB) and the 32-bit item index
    uint64_t max_frames = std::min<uint64_t>((8ull << 30) / (n_slots * sizeof(float4)), 0xffffffffull / n_slots);
    if (max_frames == 0) max_frames = 1;
    const uint32_t per_launch = (uint32_t) std::min<uint64_t>(frame_count, max_frames);
    if (ctx->stage.n < (size_t) per_launch * n_slots) HIP_TRY(ctx->stage.alloc((size_t) per_launch * n_slots));
    A.stage = ctx->stage.p;
    int per_cu = 0;
    const bool media = ctx->params.process_mediums != 0;
    const bool count = ctx->count_traversal;
    const bool deep = ctx->stack_overflow.p != nullptr;
    auto kernel = select_render_kernel(ctx->full_materials, media, count, deep);
    if (ctx->hero) HIP_TRY((ctx->hero4 ? vmk_hero4_occupancy : vmk_hero_occupancy)(ctx->full_materials, media, count, deep, &per_cu));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0));
    if (per_cu < 1) per_cu = 1;
    if (const char *cap = getenv("VMK_MAX_BLOCKS_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(cap))); // occupancy experiments (tools/gpu_occupancy.sh)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    if (!kernel_ms && ctx->timing && !ctx->in_self_check) { // asynchronous timing: events only, read back by vmk_collect_kernel_ms (the self check's own frame is not a user batch)
        if (ctx->time_used == ctx->time_pool.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
            ctx->time_pool.emplace_back(a, b);
        }
        t0 = ctx->time_pool[ctx->time_used].first; t1 = ctx->time_pool[ctx->time_used].second; ++ctx->time_used;
        HIP_TRY(hipEventRecord(t0, ctx->stream));
    }
    if (kernel_ms) HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    for (uint32_t done = 0; done < frame_count; done += per_launch) {
        A.frame_begin = frame_begin + done; A.frame_count = std::min(per_launch, frame_count - done);
        const uint64_t n_items = (uint64_t) A.frame_count * n_slots;
        A.n_items = (uint32_t) n_items;
        uint32_t grid = (uint32_t) std::min<uint64_t>((n_items + kBlock - 1) / kBlock, (uint64_t) ctx->n_cus * (uint64_t) per_cu);
        if (deep) grid = std::min<uint32_t>(grid, kOverflowWaves / (kBlock / 64));
        // With a communicator attached the persistent grid leaves a few block slots free: the exchange of the previous batch runs as RCCL
        // kernels on the second stream, and a grid that fills every wave slot would hold them back until its own blocks start to retire.
        // 1/64 of the grid (24 of 1536 blocks) costs the megakernel that fraction and lets the collective run under it.
        if (ctx->comm && ctx->comm_world > 1 && grid > 64) grid -= std::max<uint32_t>(1u, grid / 64u); // the HBM stack overflow is sized for kOverflowWaves waves of one grid
        // a wave claims `chunk` items per atomic: few enough claims to keep the counter cold, small enough to balance the tail
        uint64_t chunk = (n_items / ((uint64_t) grid * (kBlock / 64) * 8)) & ~63ull;
        A.chunk = (uint32_t) std::max<uint64_t>(64, std::min<uint64_t>(1024, chunk));
        HIP_TRY(hipMemsetAsync(ctx->queue.p, 0, sizeof(uint32_t), ctx->stream));
#ifdef VMK_DIAG
        HIP_TRY(ctx->diag.alloc((size_t) n_items * 128));
        HIP_TRY(hipMemsetAsync(ctx->diag.p, 0, (size_t) n_items * 128 * 4, ctx->stream));
        A.diag = ctx->hero ? nullptr : ctx->diag.p;
#endif
        if (ctx->hero) HIP_TRY((ctx->hero4 ? vmk_hero4_launch_render : vmk_hero_launch_render)(ctx->full_materials, media, count, deep, grid, ctx->stream, static_cast<const RenderRest *>(&A), sizeof(RenderRest), &ctx->h_scene, sizeof(DSceneFull)));
        else {
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A);
            HIP_TRY(hipGetLastError());
        }
        if (ctx->comm_pending) { HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->comm_done, 0)); ctx->comm_pending = false; } // the exchange of the previous batch reads fb
        hipLaunchKernelGGL(k_film_resolve, dim3((A.n_slots + 255) / 256), dim3(256), 0, ctx->stream, static_cast<const RenderRest &>(A));
        HIP_TRY(hipGetLastError());
    }
    if (t1) HIP_TRY(hipEventRecord(t1, ctx->stream));
    if (kernel_ms) {
        HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(hipEventSynchronize(ctx->ev1));
        HIP_TRY(hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
    }
    return VMK_OK;
}

int vmk_enable_kernel_timing(vmk_ctx *ctx, int enabled) {
    if (!ctx) return VMK_ERR_ARG;
    ctx->timing = enabled != 0;
    if (!ctx->timing) ctx->time_used = 0;
    return VMK_OK;
}
int vmk_collect_kernel_ms(vmk_ctx *ctx, float *out_ms, uint32_t max_count, uint32_t *count) {
    if (!ctx || !count || (max_count && !out_ms)) return VMK_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    uint32_t n = 0;
    for (size_t i = 0; i < ctx->time_used && n < max_count; ++i, ++n) HIP_TRY(hipEventElapsedTime(out_ms + n, ctx->time_pool[i].first, ctx->time_pool[i].second));
    *count = n;
    ctx->time_used = 0;
    return VMK_OK;
}

#ifdef VMK_DIAG
int vmk_diag_download(vmk_ctx *ctx, float *out, uint64_t n_floats) {
    if (!ctx || !out || n_floats > ctx->diag.n) return VMK_ERR_ARG;
    HIP_TRY(hipMemcpyAsync(out, ctx->diag.p, n_floats * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VMK_OK;
}
#endif
// ---- multi-GPU exchange (one collective per batch over RCCL / xGMI) ----
int vmk_comm_unique_id(void *id_out) {
    if (!id_out) return VMK_ERR_ARG;
    std::string err;
    if (!rccl_load(err)) { g_null_error = "vmk_comm_unique_id: " + err; return VMK_ERR_UNSUPPORTED; }
    static_assert(sizeof(ncclUniqueId) == VMK_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { g_null_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return VMK_ERR_HIP; }
    std::memcpy(id_out, &id, sizeof(id));
    return VMK_OK;
}
static int comm_prepare(vmk_ctx *ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->comm_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    if (!ctx->render_done) HIP_TRY(hipEventCreateWithFlags(&ctx->render_done, hipEventDisableTiming));
    if (!ctx->comm_done) HIP_TRY(hipEventCreateWithFlags(&ctx->comm_done, hipEventDisableTiming));
    return VMK_OK;
}
int vmk_comm_init(vmk_ctx *ctx, const void *unique_id, int rank, int world) {
    if (!ctx) return VMK_ERR_ARG;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) { ctx->error = "vmk_comm_init: bad argument"; return VMK_ERR_ARG; }
    if (!rccl_load(ctx->error)) return VMK_ERR_UNSUPPORTED;
    vmk_comm_release(ctx);
    int rc = comm_prepare(ctx);
    if (rc != VMK_OK) return rc;
    ncclUniqueId id; std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) { ctx->error = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); return VMK_ERR_HIP; }
    ctx->comm = comm; ctx->comm_owned = true; ctx->comm_world = world; ctx->comm_rank = rank;
    return VMK_OK;
}
int vmk_comm_adopt(vmk_ctx *ctx, void *nccl_comm) {
    if (!ctx) return VMK_ERR_ARG;
    if (!nccl_comm) { ctx->error = "vmk_comm_adopt: null communicator"; return VMK_ERR_ARG; }
    if (!rccl_load(ctx->error)) return VMK_ERR_UNSUPPORTED;
    vmk_comm_release(ctx);
    int rc = comm_prepare(ctx);
    if (rc != VMK_OK) return rc;
    int cw = 0, cr = 0; // the communicator's own idea of its size and this process's place in it (checked against every vmk_tiles handed in)
    ncclResult_t r = g_rccl.CommCount((ncclComm_t) nccl_comm, &cw);
    if (r == ncclSuccess) r = g_rccl.CommUserRank((ncclComm_t) nccl_comm, &cr);
    if (r != ncclSuccess) { ctx->error = std::string("vmk_comm_adopt: ") + g_rccl.GetErrorString(r); return VMK_ERR_HIP; }
    ctx->comm = nccl_comm; ctx->comm_owned = false; ctx->comm_world = cw; ctx->comm_rank = cr;
    return VMK_OK;
}
// enqueue "after everything rendered so far" on the exchange stream; the render stream goes on with the next batch and
// only its film resolve (the one writer of the framebuffer) waits for the exchange (vmk_render_batch)
static int comm_begin(vmk_ctx *ctx, const char *who) {
    if (!ctx->comm) { ctx->error = std::string(who) + ": no communicator (vmk_comm_init / vmk_comm_adopt)"; return VMK_ERR_STATE; }
    if (!ctx->params_ready || !ctx->fb) { ctx->error = std::string(who) + ": no framebuffer"; return VMK_ERR_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->render_done, ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ctx->render_done, 0));
    return VMK_OK;
}
int vmk_allreduce_framebuffer(vmk_ctx *ctx, void *recv_device) {
    if (!ctx) return VMK_ERR_ARG;
    if (!recv_device || recv_device == (void *) ctx->fb) { if (ctx) ctx->error = "vmk_allreduce_framebuffer: recv_device must be a second width*height*4 float buffer (the framebuffer keeps this rank's tiles only)"; return VMK_ERR_ARG; }
    int rc = comm_begin(ctx, "vmk_allreduce_framebuffer");
    if (rc != VMK_OK) return rc;
    const size_t count = (size_t) ctx->params.width * ctx->params.height * 4;
    ncclResult_t r = g_rccl.AllReduce(ctx->fb, recv_device, count, ncclFloat32, ncclSum, (ncclComm_t) ctx->comm, ctx->comm_stream);
    if (r != ncclSuccess) { ctx->error = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r); return VMK_ERR_HIP; }
    HIP_TRY(hipEventRecord(ctx->comm_done, ctx->comm_stream));
    ctx->comm_pending = true;
    return VMK_OK;
}
int vmk_allgather_framebuffer(vmk_ctx *ctx, const vmk_tiles *tiles, void *recv_device) {
    if (!ctx) return VMK_ERR_ARG;
    if (!tiles || !tiles->tile_size || (tiles->tile_size & (tiles->tile_size - 1)) || !tiles->world || tiles->rank >= tiles->world || !recv_device || recv_device == (void *) ctx->fb) { ctx->error = "vmk_allgather_framebuffer: bad argument"; return VMK_ERR_ARG; }
    if (ctx->comm && ((int) tiles->world != ctx->comm_world || (int) tiles->rank != ctx->comm_rank)) {
        // (the receive buffer and the unpack table are sized from `tiles`, ncclAllGather writes per the communicator: a mismatch is a buffer overrun)
        ctx->error = "vmk_allgather_framebuffer: vmk_tiles (rank " + std::to_string(tiles->rank) + " of " + std::to_string(tiles->world) + ") disagrees with the communicator (rank " + std::to_string(ctx->comm_rank) + " of " + std::to_string(ctx->comm_world) + ")";
        return VMK_ERR_ARG;
    }
    int rc = comm_begin(ctx, "vmk_allgather_framebuffer");
    if (rc != VMK_OK) return rc;
    const uint32_t ts = tiles->tile_size, world = tiles->world, W = ctx->params.width, H = ctx->params.height;
    const uint32_t tiles_x = (W + ts - 1) / ts, tiles_y = (H + ts - 1) / ts;
    std::vector<std::vector<uint32_t>> own(world);
    uint32_t per_rank = 0;
    for (uint32_t r = 0; r < world; ++r) { own[r] = world > 1 ? owned_tiles(tiles_x, tiles_y, r, world) : std::vector<uint32_t>(); if (world == 1) { own[0].resize((size_t) tiles_x * tiles_y); for (uint32_t t = 0; t < tiles_x * tiles_y; ++t) own[0][t] = t; } per_rank = std::max<uint32_t>(per_rank, (uint32_t) own[r].size()); }
    std::vector<uint32_t> table_all((size_t) world * per_rank, VMK_INVALID);
    for (uint32_t r = 0; r < world; ++r) std::copy(own[r].begin(), own[r].end(), table_all.begin() + (size_t) r * per_rank);
    DevBuf<uint32_t> d_table; // (small: rebuilt per call; the exchange happens once per batch of seconds)
    const size_t per_rank_px = (size_t) per_rank * ts * ts;
    hipError_t e = d_table.upload(table_all.data(), table_all.size(), ctx->comm_stream);
    if (e == hipSuccess && ctx->gather_send.n < per_rank_px) e = ctx->gather_send.alloc(per_rank_px);
    if (e == hipSuccess && ctx->gather_recv.n < per_rank_px * world) e = ctx->gather_recv.alloc(per_rank_px * world);
    if (e != hipSuccess) { d_table.release(); ctx->error = std::string("vmk_allgather_framebuffer: ") + hipGetErrorString(e); return VMK_ERR_HIP; }
    hipLaunchKernelGGL(k_pack_tiles, dim3((unsigned) ((per_rank_px + 255) / 256)), dim3(256), 0, ctx->comm_stream, ctx->fb, ctx->gather_send.p, d_table.p + (size_t) tiles->rank * per_rank, per_rank, ts, tiles_x, W, H);
    ncclResult_t r = g_rccl.AllGather(ctx->gather_send.p, ctx->gather_recv.p, per_rank_px * 4, ncclFloat32, (ncclComm_t) ctx->comm, ctx->comm_stream);
    if (r != ncclSuccess) { (void) hipStreamSynchronize(ctx->comm_stream); d_table.release(); ctx->error = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r); return VMK_ERR_HIP; }
    hipLaunchKernelGGL(k_unpack_tiles, dim3((unsigned) ((per_rank_px * world + 255) / 256)), dim3(256), 0, ctx->comm_stream, ctx->gather_recv.p, (float4 *) recv_device, d_table.p, world, per_rank, ts, tiles_x, W, H);
    e = hipEventRecord(ctx->comm_done, ctx->comm_stream);
    ctx->comm_pending = true;
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->comm_stream); // d_table is freed below (the all-reduce form stays asynchronous)
    d_table.release();
    if (e != hipSuccess) { ctx->error = std::string("vmk_allgather_framebuffer: ") + hipGetErrorString(e); return VMK_ERR_HIP; }
    return VMK_OK;
}
int vmk_comm_synchronize(vmk_ctx *ctx) {
    if (!ctx) return VMK_ERR_ARG;
    if (!ctx->comm_stream) return VMK_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->comm_stream));
    ctx->comm_pending = false;
    return VMK_OK;
}
int vmk_synchronize(vmk_ctx *ctx) {
    if (!ctx) return VMK_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VMK_OK;
}

int vmk_download_accum(vmk_ctx *ctx, float *out_rgba) {
    if (!ctx || !out_rgba) return VMK_ERR_ARG;
    if (!ctx->params_ready || !ctx->fb) { ctx->error = "vmk_download_accum: no framebuffer"; return VMK_ERR_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(out_rgba, ctx->fb, (size_t) ctx->params.width * ctx->params.height * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VMK_OK;
}

int vmk_tonemap(vmk_ctx *ctx, int final_picture, float *out_rgba) {
    if (!ctx || !out_rgba) return VMK_ERR_ARG;
    if (!ctx->params_ready || !ctx->fb) { ctx->error = "vmk_tonemap: no framebuffer"; return VMK_ERR_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    uint32_t n = ctx->params.width * ctx->params.height;
    hipLaunchKernelGGL(k_tonemap, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->fb, ctx->tm_out.p, n, ctx->params.exposure, ctx->params.tone_mapper, final_picture);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_rgba, ctx->tm_out.p, (size_t) n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VMK_OK;
}

int vmk_get_counters(vmk_ctx *ctx, vmk_counters *out) {
    if (!ctx || !out) return VMK_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned long long h[8];
    HIP_TRY(hipMemcpyAsync(h, ctx->counters.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    out->closest_rays = h[0]; out->shadow_rays = h[1]; out->nodes_visited = h[2]; out->tris_tested = h[3]; out->paths = h[4]; out->surface_hits = h[5]; out->tex_fetches = h[6];
    return VMK_OK;
}
int vmk_set_auto_self_check(vmk_ctx *ctx, int enabled) {
    if (!ctx) return VMK_ERR_ARG;
    ctx->auto_self_check = enabled != 0;
    return VMK_OK;
}

int vmk_set_traversal_counters(vmk_ctx *ctx, int enabled) {
    if (!ctx) return VMK_ERR_ARG;
    ctx->count_traversal = enabled != 0;
    return VMK_OK;
}
int vmk_reset_counters(vmk_ctx *ctx) {
    if (!ctx) return VMK_ERR_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(ctx->counters.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    return VMK_OK;
}

int vmk_trace_rays(vmk_ctx *ctx, uint32_t n, const float *org_xyz, const float *dir_xyz, const float *tmax, int any_hit, uint32_t *hit_out, float *kernel_ms, uint32_t repeats) {
    if (!ctx) return VMK_ERR_ARG;
    if (!n || !org_xyz || !dir_xyz || !tmax || !hit_out) { ctx->error = "vmk_trace_rays: bad argument"; return VMK_ERR_ARG; }
    if (!ctx->accel_ready) { ctx->error = "vmk_trace_rays: accel not built"; return VMK_ERR_STATE; }
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<float> o, d, t; DevBuf<uint32_t> h;
    auto cleanup = [&]() { o.release(); d.release(); t.release(); h.release(); };
    // host arrays are AoS (n x 3); the kernel reads SoA component planes
    std::vector<float> so((size_t) n * 3), sd((size_t) n * 3);
    for (uint32_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { so[(size_t) k * n + i] = org_xyz[(size_t) i * 3 + k]; sd[(size_t) k * n + i] = dir_xyz[(size_t) i * 3 + k]; }
    hipError_t e;
    if ((e = o.upload(so.data(), so.size(), ctx->stream)) != hipSuccess || (e = d.upload(sd.data(), sd.size(), ctx->stream)) != hipSuccess ||
        (e = t.upload(tmax, n, ctx->stream)) != hipSuccess || (e = h.alloc((size_t) n * 4)) != hipSuccess) { ctx->error = std::string("vmk_trace_rays: ") + hipGetErrorString(e); cleanup(); return VMK_ERR_HIP; }
    uint32_t grid = (uint32_t) std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t) ctx->n_cus * 6); // persistent waves, 6 blocks per CU fit (LDS)
    uint32_t chunk = (uint32_t) (n / ((uint64_t) grid * (kBlock / 64) * 2)) & ~15u; // about two claims per wave
    chunk = std::max(16u, std::min(256u, chunk));
    if (repeats == 0) repeats = 1;
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipEventRecord(ctx->ev0, ctx->stream);
    for (uint32_t r = 0; r < repeats; ++r) {
        (void) hipMemsetAsync(ctx->queue.p, 0, sizeof(uint32_t), ctx->stream);
        {
            auto kern = ctx->stack_overflow.p ? (any_hit ? k_trace<true, true> : k_trace<true, false>) : (any_hit ? k_trace<false, true> : k_trace<false, false>);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, ctx->stream, static_cast<const DScene &>(ctx->h_scene), n, o.p, d.p, t.p, any_hit, h.p, ctx->counters.p, ctx->queue.p, chunk);
        }
    }
    (void) hipEventRecord(ctx->ev1, ctx->stream);
    e = hipMemcpyAsync(hit_out, h.p, (size_t) n * 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    float ms = 0.f; (void) hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    if (kernel_ms) *kernel_ms = ms / (float) repeats;
    cleanup();
    if (e != hipSuccess) { ctx->error = std::string("vmk_trace_rays: ") + hipGetErrorString(e); return VMK_ERR_HIP; }
    return VMK_OK;
}

// general 4x4 inverse (column-major floats in, floats out) evaluated in double; false when singular
static bool inverse4(const float *m, float *out) {
    double a[16], inv[16];
    for (int i = 0; i < 16; ++i) a[i] = m[i];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0) return false;
    for (int i = 0; i < 16; ++i) out[i] = (float) (inv[i] / det);
    return true;
}

int vmk_render_aov(vmk_ctx *ctx, uint32_t frame, float *normal_rgba, float *albedo_rgba, float *emission_rgba, float *depth, float *motion_xy) {
    if (!ctx) return VMK_ERR_ARG;
    if (!ctx->accel_ready || !ctx->params_ready) { ctx->error = "vmk_render_aov: scene/accel/params not ready"; return VMK_ERR_STATE; }
    if (ctx->params.light_sampler == 1 && !ctx->has_light_alias) { ctx->error = "vmk_render_aov: the power light sampler needs vmk_scene::light_alias_offset"; return VMK_ERR_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t) ctx->params.width * ctx->params.height;
    DevBuf<float4> dn, da, de; DevBuf<float> dd; DevBuf<float2> dm;
    auto cleanup = [&]() { dn.release(); da.release(); de.release(); dd.release(); dm.release(); };
    hipError_t e = hipSuccess;
    if (normal_rgba) e = dn.alloc(n);
    if (e == hipSuccess && albedo_rgba) e = da.alloc(n);
    if (e == hipSuccess && emission_rgba) e = de.alloc(n);
    if (e == hipSuccess && depth) e = dd.alloc(n);
    if (e == hipSuccess && motion_xy) e = dm.alloc(n);
    if (e != hipSuccess) { cleanup(); ctx->error = std::string("vmk_render_aov: ") + hipGetErrorString(e); return VMK_ERR_HIP; }
    AovArgs A{};
    A.scene = ctx->d_scene.p; A.params = ctx->d_params.p; A.normal = dn.p; A.albedo = da.p; A.emission = de.p; A.depth = dd.p; A.motion = dm.p; A.frame = frame;
    if (!inverse4(ctx->params.c2w, A.w2s) || !inverse4(ctx->params.raster_to_sensor, A.s2r)) { cleanup(); ctx->error = "vmk_render_aov: camera matrix is singular"; return VMK_ERR_ARG; }
    uint32_t grid = (uint32_t) std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t) ctx->n_cus * 6);
    if (ctx->hero) e = (ctx->hero4 ? vmk_hero4_launch_aov : vmk_hero_launch_aov)(grid, ctx->stream, &A, sizeof(A)); // (same AovArgs layout in every instance)
    else {
        hipLaunchKernelGGL(k_aov, dim3(grid), dim3(kBlock), 0, ctx->stream, A);
        e = hipGetLastError();
    }
    if (e == hipSuccess && normal_rgba) e = hipMemcpyAsync(normal_rgba, dn.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && albedo_rgba) e = hipMemcpyAsync(albedo_rgba, da.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && emission_rgba) e = hipMemcpyAsync(emission_rgba, de.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && depth) e = hipMemcpyAsync(depth, dd.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && motion_xy) e = hipMemcpyAsync(motion_xy, dm.p, n * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) { ctx->error = std::string("vmk_render_aov: ") + hipGetErrorString(e); return VMK_ERR_HIP; }
    return VMK_OK;
}

int vmk_precompute_albedo(vmk_ctx *ctx, uint32_t which, uint32_t res, uint32_t sample_num, float *out) {
    if (!ctx) return VMK_ERR_ARG;
    if (which > 4 || res < 2 || res > 128 || sample_num == 0 || !out) { ctx->error = "vmk_precompute_albedo: bad argument"; return VMK_ERR_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t texels = (size_t) res * res * (which == 0 ? 1 : res), n = texels * ((which == 1 || which == 2) ? 2 : 1);
    DevBuf<float> d;
    hipError_t e = d.alloc(n);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_albedo, dim3((unsigned) ((texels + 63) / 64)), dim3(64), 0, ctx->stream, which, res, sample_num, d.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    d.release();
    if (e != hipSuccess) { ctx->error = std::string("vmk_precompute_albedo: ") + hipGetErrorString(e); return VMK_ERR_HIP; }
    return VMK_OK;
}

int vmk_test_eval(vmk_ctx *ctx, uint32_t kind, uint32_t n, const float *in, uint32_t in_stride, float *out, uint32_t out_stride) {
    if (!ctx) return VMK_ERR_ARG;
#ifdef VMK_DIAG
    const uint32_t kind_arg = kind;
    if (kind == 8) kind = 6;
#endif
    if (!n || !in || !out || !in_stride || !out_stride || kind > 7) { ctx->error = "vmk_test_eval: bad argument"; return VMK_ERR_ARG; }
    static const uint32_t min_in[8] = {4, 2, 2, 8, 12, 3, 3, 3}, min_out[8] = {8, 6, 8, 8, 13, 6, 67, 1 + 16 * 24};
    if (in_stride < min_in[kind] || out_stride < min_out[kind]) { ctx->error = "vmk_test_eval: stride too small for this kind"; return VMK_ERR_ARG; }
    if (kind == 4 && !ctx->accel_ready) { ctx->error = "vmk_test_eval: kind 4 needs an uploaded scene + accel"; return VMK_ERR_STATE; }
    if ((kind == 6 || kind == 7) && (!ctx->accel_ready || !ctx->params_ready)) { ctx->error = "vmk_test_eval: kinds 6/7 need scene, accel and render params"; return VMK_ERR_STATE; }
    if ((kind == 6 || kind == 7) && ctx->stack_overflow.p && (n + 63) / 64 > kOverflowWaves) { ctx->error = "vmk_test_eval: too many paths for one launch on a scene whose tree uses the HBM stack overflow"; return VMK_ERR_ARG; }
    if ((kind == 6 || kind == 7) && ctx->params.light_sampler == 1 && !ctx->has_light_alias) { ctx->error = "vmk_test_eval: the power light sampler needs vmk_scene::light_alias_offset"; return VMK_ERR_ARG; }
    if ((kind == 6 || kind == 7) && ctx->params.process_mediums && ctx->params.camera_medium != VMK_INVALID && ctx->params.camera_medium >= ctx->n_mediums) { ctx->error = "vmk_test_eval: camera medium out of range"; return VMK_ERR_ARG; }
    if (kind == 5 && !ctx->params_ready) { ctx->error = "vmk_test_eval: kind 5 needs render params"; return VMK_ERR_STATE; }
    if ((kind == 4 || kind == 7) && ctx->hero) { ctx->error = "vmk_test_eval: the material unit kernel and the ray capture are the sRGB instance; a hero-spectrum scene is uploaded (kind 6, the path unit kernel, exists for both)"; return VMK_ERR_UNSUPPORTED; }
    if (kind == 4) { // material ids are validated here: the kernel indexes materials[] with them
        uint32_t n_mat = (uint32_t) ctx->materials.n;
        for (uint32_t i = 0; i < n; ++i) { uint32_t id; std::memcpy(&id, in + (size_t) i * in_stride, 4); if (id >= n_mat) { ctx->error = "vmk_test_eval: material id out of range"; return VMK_ERR_ARG; } }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<float> di, dout;
    hipError_t e = di.upload(in, (size_t) n * in_stride, ctx->stream);
    if (e == hipSuccess) e = dout.alloc((size_t) n * out_stride);
    if (e == hipSuccess) e = hipMemsetAsync(dout.p, 0, (size_t) n * out_stride * 4, ctx->stream);
    if (e == hipSuccess) {
#ifdef VMK_DIAG
        kind = kind_arg;
#endif
        if (ctx->hero && kind == 6) e = (ctx->hero4 ? vmk_hero4_launch_unit_path : vmk_hero_launch_unit_path)(ctx->stream, ctx->d_scene.p, ctx->d_params.p, n, di.p, in_stride, dout.p, out_stride);
        else {
            hipLaunchKernelGGL(k_test, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, ctx->d_scene.p, ctx->d_params.p, kind, n, di.p, in_stride, dout.p, out_stride);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, (size_t) n * out_stride * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    di.release(); dout.release();
    if (e != hipSuccess) { ctx->error = std::string("vmk_test_eval: ") + hipGetErrorString(e); return VMK_ERR_HIP; }
    return VMK_OK;
}

// Cross-check of two independently compiled instances of the path code: frame 0 of a strided subset of pixels is rendered
// by the megakernel variant this scene selects (k_render<FULL, MEDIA>, its own register budget) and stepped by k_test kind 6
// (path_bounce<true, true> at 128 registers); the radiance must agree bit for bit.  A difference means the toolchain
// produced inconsistent code for one of them (seen twice during development on 96-register variants).
int vmk_self_check(vmk_ctx *ctx, uint32_t max_pixels, uint32_t *n_checked, uint32_t *n_mismatch) {
    if (!ctx) return VMK_ERR_ARG;
    if (!ctx->accel_ready || !ctx->params_ready || !ctx->fb) { ctx->error = "vmk_self_check: scene/accel/params not ready"; return VMK_ERR_STATE; }
    if (n_checked) *n_checked = 0;
    if (n_mismatch) *n_mismatch = 0;
    const uint32_t w = ctx->params.width, h = ctx->params.height;
    const size_t n_pix = (size_t) w * h;
    if (max_pixels == 0) max_pixels = 4096;
    HIP_TRY(hipSetDevice(ctx->device));
    // 1. frame 0 through the megakernel into a scratch film (acc = lerp(1, 0, L) = L), counters preserved
    DevBuf<float4> film;
    unsigned long long saved[8];
    HIP_TRY(hipMemcpyAsync(saved, ctx->counters.p, sizeof(saved), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipError_t e = film.alloc(n_pix);
    if (e != hipSuccess) { ctx->error = std::string("vmk_self_check: ") + hipGetErrorString(e); return VMK_ERR_HIP; }
    float4 *user_fb = ctx->fb;
    ctx->fb = film.p;
    int rc = vmk_reset_accum(ctx);
    if (rc == VMK_OK) rc = vmk_render_batch(ctx, 0, 1, nullptr, nullptr);
    std::vector<float> img(n_pix * 4);
    if (rc == VMK_OK) rc = vmk_download_accum(ctx, img.data());
    ctx->fb = user_fb;
    film.release();
    (void) hipMemcpyAsync(ctx->counters.p, saved, sizeof(saved), hipMemcpyHostToDevice, ctx->stream);
    (void) hipStreamSynchronize(ctx->stream);
    if (rc != VMK_OK) return rc;
    // 2. the same paths through the unit kernel
    const size_t stride = std::max<size_t>(1, (n_pix + max_pixels - 1) / max_pixels);
    std::vector<uint32_t> idx;
    for (size_t p = 0; p < n_pix; p += stride) idx.push_back((uint32_t) p);
    std::vector<float> in(idx.size() * 3), out(idx.size() * 67);
    for (size_t k = 0; k < idx.size(); ++k) {
        uint32_t v[3] = {idx[k] % w, idx[k] / w, 0u};
        std::memcpy(&in[k * 3], v, 12);
    }
    rc = vmk_test_eval(ctx, 6, (uint32_t) idx.size(), in.data(), 3, out.data(), 67);
    (void) hipMemcpyAsync(ctx->counters.p, saved, sizeof(saved), hipMemcpyHostToDevice, ctx->stream);
    (void) hipStreamSynchronize(ctx->stream);
    if (rc != VMK_OK) return rc;
    uint32_t bad = 0;
    for (size_t k = 0; k < idx.size(); ++k)
        for (int c = 0; c < 3; ++c) {
            float a = img[(size_t) idx[k] * 4 + c], b = out[k * 67 + 64 + c];
            uint32_t ua, ub; std::memcpy(&ua, &a, 4); std::memcpy(&ub, &b, 4);
            if (ua != ub && !(a != a && b != b)) { ++bad; break; }
        }
    if (n_checked) *n_checked = (uint32_t) idx.size();
    if (n_mismatch) *n_mismatch = bad;
    if (!bad) ctx->self_checked = true;
    if (bad) { ctx->error = "vmk_self_check: " + std::to_string(bad) + " of " + std::to_string(idx.size()) + " pixels differ between the megakernel and the unit kernel"; return VMK_ERR_STATE; }
    return VMK_OK;
}

}// extern "C"
