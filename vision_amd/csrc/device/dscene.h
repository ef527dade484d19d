// dscene.h — device-resident scene view + BVH node layout (HBM data layout, see DESIGN.md §"data layout in HBM").
#pragma once
#include "../../../include/vmk.h"
#include "dmath.h"

namespace vmkd {

typedef float f2v __attribute__((ext_vector_type(2)));

// BVH2 node, 64 B = one S_node record: both child AABBs live in the parent so one fetch decides both
// children.  Each slab is a (min, max) pair — the operand shape of the packed-fp32 slab test in dbvh.h.  child >= 0: internal node index.  child < 0: leaf, v = ~child, first triangle = v & 0x0fffffff (index
// into the Morton-ordered triangle arrays), count = (v >> 28) + 1.
struct alignas(16) BvhNode {
    float lx[2], ly[2], lz[2]; // left child:  (min, max) per axis
    float rx[2], ry[2], rz[2]; // right child
    int32_t left, right;
    uint32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 B");
constexpr uint32_t kLeafFirstMask = 0x0fffffffu;
constexpr int kMaxLeafTris = 4;

struct DScene {
    const vmk_tri_pos *tri_pos;   // Morton (BVH leaf) order, 48 B each
    const vmk_tri_attr *tri_attr; // same order, 64 B each
    const uint32_t *tri_lookup;   // instance-order global triangle index -> BVH-order index
    const vmk_instance *instances;
    const vmk_material *materials;
    const vmk_light *lights;
    const vmk_texture *textures;
    const uint8_t *tex_data;
    const float *alias_prob;
    const uint32_t *alias_idx;
    const float *alias_func;
    const float *srgb_lut; // 256-entry sRGB EOTF table (8-bit texel -> linear)
    const float *lut_pure_reflection, *lut_dielectric, *lut_dielectric_inv, *lut_specular, *lut_coat, *lut_sheen_approx;
    const BvhNode *nodes;
    int32_t root; // child-encoded reference of the root (leaf-encoded when the scene has <= kMaxLeafTris triangles)
    uint32_t n_tris, n_lights, env_light;
};

struct DCounters { // per-lane tallies, wave-reduced into vmk_counters at kernel end
    uint32_t closest, shadow, nodes, tris, paths, hits, tex;
};

}// namespace vmkd
