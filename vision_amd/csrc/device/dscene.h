// dscene.h — device-resident scene view + BVH node layout (HBM data layout, see DESIGN.md §"data layout in HBM").
#pragma once
#include "../../../include/vmk.h"
#include "dmath.h"

namespace vmkd {

typedef float f2v __attribute__((ext_vector_type(2)));

// BVH4 node, 128 B = one cache line = one S_node record.  Child-major: child c occupies bytes
// [32c, 32c+32) so that lane c of a traversal quad fetches its child with two 16 B loads and the quad's 4 lanes cover the
// line.  Each slab is a (min, max) pair — the operand shape of the packed-fp32 slab test in dbvh.h.
// ref >= 0: internal node index.  ref == kEmptyRef: unused slot.  ref < 0: leaf, v = ~ref, first triangle = v & 0x0fffffff
// (index into the depth-first ordered triangle arrays), count = (v >> 28) + 1 (<= kMaxLeafTris = one triangle per lane).
struct alignas(16) BvhChild {
    float x[2], y[2], z[2];
    int32_t ref;
    uint32_t pad;
};
struct alignas(128) BvhNode { BvhChild child[4]; };
static_assert(sizeof(BvhNode) == 128, "BvhNode must be 128 B");
constexpr int32_t kEmptyRef = 0x7ffffffe;
constexpr uint32_t kLeafFirstMask = 0x0fffffffu;
constexpr int kMaxLeafTris = 4;

struct DSceneBase {
    const vmk_tri_pos *tri_pos;   // Morton (BVH leaf) order, 48 B each
    const vmk_tri_attr *tri_attr; // same order, 64 B each
    const uint32_t *tri_lookup;   // instance-order global triangle index -> BVH-order index
    const vmk_instance *instances;
    const vmk_material *materials;
    const vmk_light *lights;
    const vmk_medium *mediums;
    const vmk_texture *textures;
    const uint8_t *tex_data;
    const float *alias_prob;
    const uint32_t *alias_idx;
    const float *alias_func;
    const float *srgb_lut; // 256-entry sRGB EOTF table (8-bit texel -> linear)
    const float *lut_pure_reflection, *lut_dielectric, *lut_dielectric_inv, *lut_specular, *lut_coat, *lut_sheen_approx;
    const BvhNode *nodes;
    int32_t root; // reference of the root: node 0, or leaf-encoded when the scene has <= kMaxLeafTris triangles
    uint32_t n_tris, n_lights, env_light;
    uint32_t light_alias_offset; // lightsampler/power table (VMK_INVALID when absent)
    float light_alias_integral;
    // Deep trees only (worst-case traversal stack need > the LDS stack, dbvh.h kQuadStack): HBM overflow of the per-ray stacks,
    // [wave of the grid][entry beyond the LDS stack][lane]; null otherwise.
    uint32_t *stack_overflow;
};
// hero spectrum only: sRGB uplift table float4[3][64][64][64], tabulated-spectra pool, CIE x, y, z, D65 (vmk_scene)
struct DSceneHero {
    const float *rgb2spec;
    const float *spd_data;
    uint32_t spd_cie[4];
    float spd_cie_interval, cie_y_integral;
};
// What the host uploads.  The sRGB instance of the kernels reads the DSceneBase prefix only (its per-lane copy of the
// scene view stays as small as before); the hero instance (VMK_HERO, vmk_hero.hip) sees the tail as well.
struct DSceneFull : DSceneBase { DSceneHero hero; };
#if defined(VMK_HERO) && VMK_HERO
using DScene = DSceneFull;
#else
using DScene = DSceneBase;
#endif

// Every pointer of the scene view comes out of memory, so the compiler has to treat it as a generic ("flat") address:
// flat_load with 64-bit VALU address arithmetic, counted on both vmcnt and lgkmcnt (it then serialises against the LDS
// traffic of the traversal).  They all point into hipMalloc'ed global memory; ldg() says so at the access.
template<typename T>
VD T ldg(const T *p) { return *(const __attribute__((address_space(1))) T *) p; } // scalars
typedef float f4v __attribute__((ext_vector_type(4)));
VD float4 ldg(const float4 *p) { f4v v = *(const __attribute__((address_space(1))) f4v *) p; return make_float4(v.x, v.y, v.z, v.w); }
// 16 B at base + a 32-bit byte offset: the shape the global_load "scalar base + 32-bit VGPR offset" form wants
VD float4 ldg_off(const void *base, uint32_t byte_off) {
    const __attribute__((address_space(1))) char *b = (const __attribute__((address_space(1))) char *) base;
    f4v v = *(const __attribute__((address_space(1))) f4v *) (b + byte_off);
    return make_float4(v.x, v.y, v.z, v.w);
}

struct DCounters { // tallies, reduced into vmk_counters at kernel end: closest / shadow / paths / hits are WAVE totals (updated with one
                   // scalar add per wave from a ballot, so they live in scalar registers), nodes / tris / tex are per lane
    uint32_t closest, shadow, nodes, tris, paths, hits, tex;
};

}// namespace vmkd
