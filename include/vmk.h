/*
 * vmk.h — C-ABI of the MI355X-native megakernel path-tracing backend ("vmk") that sits behind
 * Vision's Integrator/Pipeline plugin surface.
 *
 * Everything below Vision's `Integrator::render()` and the scene-upload half of `Pipeline::prepare()`
 * is replaced by this library.  Reference interfaces each entry point stands in for (paths relative to
 * the Vision source tree, file:line):
 *
 *   vmk_create / vmk_destroy        RHIContext::create_device + Pipeline ctor   src/apps/vision-gui/application.h:64-69,
 *                                                                               src/base/mgr/pipeline.cpp:13-31
 *   vmk_upload_scene                Scene::prepare + Geometry::update_instances/upload + ImagePool::prepare +
 *                                   upload_bindless_array                       src/base/mgr/scene.cpp:79-91,
 *                                                                               src/base/mgr/geometry.cpp:20-34,64-71,
 *                                                                               src/base/mgr/image_pool.cpp:48-54
 *   vmk_build_accel                 Geometry::build_accel (OptiX BLAS/TLAS)     src/base/mgr/geometry.cpp:36-53
 *   vmk_set_render_params           Sensor::update_device_data, IlluminationIntegrator ctor params,
 *                                   FrameBuffer ctor                            src/base/sensor/sensor.cpp:141-145,
 *                                                                               src/base/integral/integrator.cpp:59-66,
 *                                                                               src/base/sensor/frame_buffer.cpp:15-26
 *   vmk_render_batch                PathTracingIntegrator::render() x frame_count (rt_geom ray-gen + path_tracing +
 *                                   accumulate, fused)                          src/render_core/integrator/pt.cpp:96-116,
 *                                                                               src/base/integral/integrator.cpp:82-107,
 *                                                                               src/base/sensor/frame_buffer.cpp:117-126,156-219
 *   vmk_reset_accum                 Integrator::invalidation()                  src/base/integral/integrator.cpp:13-18
 *   vmk_tonemap                     FrameBuffer tone_mapping + gamma kernels,
 *                                   Pipeline::final_picture                     src/base/sensor/frame_buffer.cpp:135-154,
 *                                                                               src/base/mgr/pipeline.cpp:337-354
 *   vmk_download_accum              FrameBuffer download                        src/base/sensor/frame_buffer.cpp:424-427,467-469
 *   vmk_last_error                  OC_ERROR logging (reference has no error returns; src/base/node.h:116)
 *
 * Conventions: plain C, no C++/torch types.  Every function returns 0 on success and a negative
 * vmk_status on failure; vmk_last_error(ctx) gives the message.  A ctx is bound to ONE GPU and is not
 * thread-safe; drive one ctx per GPU (one process per GPU under torch.distributed).  The creator owns
 * every handle; host tables passed to vmk_upload_scene are copied and may be freed afterwards.
 * All matrices are column-major float[16]/float[9] like ocarina's float4x4 (m[col*4+row]).
 */
#ifndef VMK_H
#define VMK_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VMK_ABI_VERSION 7u
#define VMK_INVALID 0xFFFFFFFFu

typedef enum vmk_status {
    VMK_OK = 0,
    VMK_ERR_ARG = -1,        /* bad argument / inconsistent table sizes */
    VMK_ERR_HIP = -2,        /* a HIP runtime call failed               */
    VMK_ERR_STATE = -3,      /* call order violated (e.g. render before build_accel) */
    VMK_ERR_UNSUPPORTED = -4 /* scene uses a feature outside the hot-path scope */
} vmk_status;

/* ---- material / light vocabulary (Vision plugin type names, src/render_core/material, light) ---- */
typedef enum vmk_material_type {
    VMK_MAT_DIFFUSE = 0,    /* "diffuse"          diffuse.cpp:21-30   slots: color, sigma */
    VMK_MAT_MIRROR = 1,     /* "mirror"           mirror.cpp:60-74    slots: color, roughness, anisotropic */
    VMK_MAT_METAL = 2,      /* "metal"            metal.cpp:137-156   slots: eta, k, roughness, anisotropic */
    VMK_MAT_GLASS = 3,      /* "glass"            glass.cpp:240-257   slots: color, ior, roughness, anisotropic */
    VMK_MAT_SUBSTRATE = 4,  /* "substrate"        substrate.cpp:126-149 slots: color, spec, roughness, anisotropic */
    VMK_MAT_PRINCIPLED = 5, /* "principled_bsdf"  principled_bsdf.cpp:352-461, 18 slots in declaration order */
    VMK_MAT_MIX = 6,        /* "mix"              mix.cpp:66-71       slot: frac; children child0/child1 */
    VMK_MAT_METALLIC = 7,   /* "metallic"         metallic.cpp:24-60  slots: color, edge_tint, roughness, anisotropic (F82-tint conductor) */
    VMK_MAT_ADD = 8,        /* "add"              add.cpp:9-60        children child0/child1, LobeSet::create_add (lobe.cpp:510-522) */
    VMK_MAT_PLASTIC = 9     /* "plastic"          plastic.cpp:11-131  slots: color, spec, ior, roughness, anisotropic */
} vmk_material_type;
/* single-lobe types may be children of mix / add */
#define VMK_MAT_IS_SINGLE_LOBE(t) ((t) <= VMK_MAT_SUBSTRATE || (t) == VMK_MAT_METALLIC || (t) == VMK_MAT_PLASTIC)

enum { /* principled slot indices, principled_bsdf.cpp:235-256 */
    VMK_P_COLOR = 0, VMK_P_METALLIC, VMK_P_IOR, VMK_P_ROUGHNESS, VMK_P_SPEC_TINT, VMK_P_ANISOTROPIC, VMK_P_OPACITY,
    VMK_P_SHEEN_WEIGHT, VMK_P_SHEEN_ROUGHNESS, VMK_P_SHEEN_TINT, VMK_P_COAT_WEIGHT, VMK_P_COAT_ROUGHNESS,
    VMK_P_COAT_IOR, VMK_P_COAT_TINT, VMK_P_SSS_WEIGHT, VMK_P_SSS_RADIUS, VMK_P_SSS_SCALE, VMK_P_TRANS_WEIGHT,
    VMK_P_SLOT_COUNT
};
#define VMK_MAX_SLOTS 18

#define VMK_MATF_REMAP_ROUGHNESS 1u /* desc["remapping_roughness"] (default true) */
#define VMK_MATF_HAS_SIGMA 2u       /* diffuse: Oren-Nayar when "sigma" present (diffuse.cpp:24-27) */
#define VMK_MATF_DISPERSIVE 4u      /* glass, hero spectrum: the ior slot is an "spd" node (GlassMaterial::is_dispersive glass.cpp:234) */
#define VMK_MATF_HAS_NORMAL 8u      /* desc.has_attr("normal") (material.cpp:312-316): vmk_material::normal feeds compute_shading_frame */

/* A material / light parameter slot (ShaderNodeSlot, src/base/shader_graph/shader_node.cpp:242-273).
 * tex == VMK_INVALID : constant, value v[0..2] (scalar slots use v[0]).
 * otherwise          : image node (render_core/shadernode/image.cpp:87-97): texel(uv) * v[0](=scale),
 *                      low 16 bits = texture index, bits 16..21 = 3 x 2-bit channel swizzle (x,y,z sources).
 *                      bit 22 (VMK_SLOT_TINTED): a "multiply" node of that image with a constant (render_core/shadernode/math.cpp:34-92,
 *                      BinaryOpNode): swizzle(texel(uv)) * (v[0], v[1], v[2]) per channel; the image's own scale is 1 then. */
#define VMK_SLOT_TINTED (1u << 22)
typedef struct vmk_slot {
    float v[3];
    uint32_t tex;
} vmk_slot;
/* hero spectrum only — "spd" shader node (render_core/shadernode/spd.cpp:36-39): a tabulated spectrum evaluated at the
 * path's sampled wavelengths.  tex == VMK_SLOT_SPD, v[0] / v[1] hold the BIT PATTERNS of two uint32 (first float of the
 * table in vmk_scene.spd_data, sample count), v[2] = SPD::sample_interval_ (spd.cpp:50-53: 471 / count). */
#define VMK_SLOT_SPD 0xFFFFFFFDu

/* Spectrum plugin (src/render_core/spectrum): how colours travel along a path */
typedef enum vmk_spectrum_type {
    VMK_SPECTRUM_SRGB = 0, /* srgb.cpp: three fixed channels, RGB values used as they are */
    VMK_SPECTRUM_HERO = 1  /* hero.cpp: 3 or 4 wavelengths per path (hero + rotations, vmk_scene::spectrum_dimension), RGB uplifted through the sigmoid table */
} vmk_spectrum_type;
#define VMK_RGB2SPEC_RES 64u /* RGBToSpectrumTable::res hero.cpp:53 */

typedef struct vmk_material {
    uint32_t type;  /* vmk_material_type */
    uint32_t flags; /* VMK_MATF_* */
    uint32_t child0, child1; /* mix / add only: indices into materials[]: single-lobe types, or ONE principled_bsdf next to a
                              * single-lobe type (LobeSet::flatten lobe.cpp:534-562 merges its lobes into the parent's list) */
    vmk_slot slot[VMK_MAX_SLOTS];
    vmk_slot normal; /* VMK_MATF_HAS_NORMAL: tangent-space normal (Material::compute_shading_frame material.cpp:331-353) */
} vmk_material;

typedef enum vmk_light_type {
    VMK_LIGHT_AREA = 0,     /* "area"      render_core/light/area.cpp */
    VMK_LIGHT_SPHERICAL = 1, /* "spherical" render_core/light/environments/spherical.cpp */
    VMK_LIGHT_POINT = 2,    /* "point"     render_core/light/point.cpp:19-48; IPointLight light.h:222-251, light.cpp:49-58 */
    VMK_LIGHT_SPOT = 3,     /* "spot"      render_core/light/spot.cpp:19-80 */
    VMK_LIGHT_PROJECTOR = 4 /* "projector" render_core/light/projector.cpp:31-111: a point light whose colour is an image seen through a frustum */
} vmk_light_type;

typedef struct vmk_light {
    uint32_t type;
    uint32_t inst_id;      /* area: emissive instance */
    uint32_t two_sided;    /* area */
    float scale;           /* Light::scale_ after color normalisation (light.cpp:19-24) */
    vmk_slot color;        /* normalised colour (max component <= 1) or image */
    uint32_t alias_offset; /* area: 1-D alias table over the instance's triangles (by area); env: marginal (rows) */
    uint32_t alias_count;
    float alias_integral;  /* AliasTable::integral_ = sum/size (alias.h:119) */
    uint32_t cond_offset;  /* env: flat conditional tables, res.x entries per row (alias2d.cpp:56-66) */
    uint32_t res_x, res_y; /* env: importance-map resolution */
    float w2o[9];          /* env: 3x3 of w2o_  (spherical.cpp:36-44), column-major */
    float o2w[9];          /* env: 3x3 of inverse(w2o_) as evaluated by the reference per sample (spherical.cpp:114) */
    float world_diameter;  /* env: Scene::world_diameter() (scene.h:108-109) */
    float position[3];     /* point / spot */
    float direction[3];    /* spot: normalised axis */
    float cos_angle;       /* spot: cos(angle), angle clamped to [1, 89] degrees (spot.cpp:30) */
    float cos_falloff_start; /* spot: cos(max(0, angle - falloff)) (spot.cpp:57-59) */
    float w2o4[16];        /* projector: inverse(o2w_) as Projector::Le evaluates it per sample (projector.cpp:100), column-major 4x4;
                            * position[] = o2w_[3].xyz */
    float tan_xy[2];       /* projector: (ratio * tan(angle_y), tan(angle_y)), angle clamped to [1, 89] degrees (:41-47,104-106) */
} vmk_light;

/* ---- geometry -------------------------------------------------------------------------------- */
/* One record per triangle, instance order (instance i owns [tri_offset, tri_offset+tri_count)).
 * Positions are WORLD space: p = o2w.apply_point(v.position) evaluated on the host with the same
 * float arithmetic compute_surface_interaction uses per hit (geometry.cpp:94-96). 48 B = S_tri. */
typedef struct vmk_tri_pos {
    float p0[3], p1[3], p2[3];
    uint32_t inst; /* instance index */
    uint32_t prim; /* triangle index inside the instance's mesh (TriangleHit::prim_id) */
    uint32_t pad;
} vmk_tri_pos;

/* Shading attributes of the same triangle: OBJECT-space vertex normals (transformed after
 * interpolation like geometry.cpp:128-133) and texture coordinates. 64 B. */
typedef struct vmk_tri_attr {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    float pad;
} vmk_tri_attr;

typedef struct vmk_instance { /* InstanceData, src/base/shape.h:21-33 */
    uint32_t mat_id;   /* index into materials[] or VMK_INVALID */
    uint32_t light_id; /* index into lights[] or VMK_INVALID */
    uint32_t tri_offset, tri_count;
    float n2w[9];      /* normal matrix: transpose(inverse(o2w 3x3)), column-major (Transform::apply_normal) */
    float o2w[16];     /* kept for reference / debugging */
    uint32_t inside_medium, outside_medium; /* indices into mediums[] or VMK_INVALID (shape.cpp:246-271, scene.cpp:135-147) */
} vmk_instance;

/* ---- participating media (render_core/medium/homogeneous.cpp:11-70) ------------------------------------ */
typedef struct vmk_medium { /* "homogeneous": sigma_t = (sigma_a + sigma_s) * scale, Henyey-Greenstein g clamped to +-0.99 */
    float sigma_a[3], sigma_s[3];
    float g, scale;
} vmk_medium;

/* ---- textures ----------------------------------------------------------------------------------- */
typedef enum vmk_tex_format { VMK_TEX_RGBA8_SRGB = 0, VMK_TEX_RGBA8_LINEAR = 1, VMK_TEX_RGBA32F = 2 } vmk_tex_format;
typedef struct vmk_texture {
    uint64_t offset; /* byte offset into tex_data (16-byte aligned) */
    uint32_t width, height;
    uint32_t format; /* vmk_tex_format */
    uint32_t channels; /* channel count of the source image (ImageNode::channel_num) */
} vmk_texture;

/* ---- precomputed albedo tables (base/scattering/precomputed_table.h, ltc_sheen_table.h) -------------- */
#define VMK_LUT_RES 32
typedef struct vmk_luts {
    const float *pure_reflection; /* [32*32]      PureReflectionLobe::lut   (x=alpha, y=cos) */
    const float *dielectric;      /* [32^3 * 2]   DielectricLobe::lut       float2 */
    const float *dielectric_inv;  /* [32^3 * 2]   DielectricLobeInv::lut    float2 */
    const float *specular;        /* [32^3]       SpecularLobe::lut */
    const float *coat;            /* [32^3]       CoatLobe::lut */
    const float *sheen_approx;    /* [32*32*4]    SheenLTC::Approximate (may be NULL -> sheen disabled) */
    const float *sheen_volume;    /* [32*32*4]    SheenLTC::Volume */
} vmk_luts;

/* ---- whole scene ------------------------------------------------------------------------------------ */
typedef struct vmk_scene {
    uint32_t abi_version;
    uint32_t n_tris, n_instances, n_materials, n_lights, n_textures, n_alias;
    const vmk_tri_pos *tri_pos;
    const vmk_tri_attr *tri_attr;
    const vmk_instance *instances;
    const vmk_material *materials;
    const vmk_light *lights;     /* order = LightSampler order after tidy_up (lightsampler.cpp:64-74) */
    const vmk_texture *textures;
    const uint8_t *tex_data;
    uint64_t tex_bytes;
    const float *alias_prob;     /* AliasEntry::prob   (alias.h:13-16) */
    const uint32_t *alias_idx;   /* AliasEntry::alias */
    const float *alias_func;     /* AliasTable::func_ */
    uint32_t env_light;          /* index of the environment light in lights[] or VMK_INVALID */
    float world_min[3], world_max[3];
    vmk_luts luts;
    uint32_t n_mediums;
    const vmk_medium *mediums; /* MediumRegistry order (scene.cpp:189-199) */
    /* lightsampler/power (render_core/lightsampler/power.cpp:35-52): alias table over luminance(light->power()), one
     * entry per light in lights[] order (the environment light weighs 0 when env_separate); VMK_INVALID when absent */
    uint32_t light_alias_offset;
    float light_alias_integral;
    /* ---- spectrum (render_core/spectrum/{srgb,hero}.cpp); everything below is ignored for VMK_SPECTRUM_SRGB ---- */
    uint32_t spectrum;          /* vmk_spectrum_type */
    const float *rgb2spec;      /* float[3][64][64][64][4]: sigmoid-polynomial coefficients c0,c1,c2 (w unused), hero.cpp:52-76 */
    const float *spd_data;      /* pool of tabulated spectra: the four CIE tables below + every VMK_SLOT_SPD slot */
    uint32_t n_spd;             /* floats in spd_data */
    uint32_t spd_cie[4];        /* first float of CIE x, y, z and illuminant D65 (SPD::create_cie_* spd.cpp:95-109) */
    uint32_t spd_cie_count;     /* samples per CIE table (94: every 5th of the 1 nm tables) */
    float spd_cie_interval;     /* SPD::sample_interval_ of those tables (471 / 94) */
    float cie_y_integral;       /* SPD::cie_y_integral() spd.cpp:111-114 */
    uint32_t spectrum_dimension; /* HeroWavelengthSpectrum::dimension_ (hero.cpp:229,240): 3 (also 0) or 4; each is its own megakernel instance */
} vmk_scene;

/* ---- camera / film / integrator ---------------------------------------------------------------------- */
typedef enum vmk_filter_type { VMK_FILTER_BOX = 0, VMK_FILTER_TRIANGLE = 1, VMK_FILTER_TABLE = 2 } vmk_filter_type;
#define VMK_FILTER_TABLE_SIZE 20 /* FilterSampler::table_size, fitted_curve.h:19 */

typedef struct vmk_render_params {
    uint32_t width, height;
    /* Sensor (sensor.cpp:44-71,153-162; thin_lens.cpp:34-42) */
    float c2w[16];              /* camera_to_world() */
    float raster_to_sensor[16]; /* inverse(perspective) * raster_to_screen */
    float lens_radius, focal_distance;
    /* Filter (box.cpp:16-20, triangle.cpp:16-18, fitted_curve.h:76-113) */
    uint32_t filter_type;
    float filter_radius[2];
    /* fitted-curve filters (gaussian/mitchell/sinc): alias-2D over |f| on a 20x20 grid */
    float filter_marginal_prob[VMK_FILTER_TABLE_SIZE];
    uint32_t filter_marginal_alias[VMK_FILTER_TABLE_SIZE];
    float filter_marginal_func[VMK_FILTER_TABLE_SIZE];
    float filter_marginal_integral;
    float filter_cond_prob[VMK_FILTER_TABLE_SIZE * VMK_FILTER_TABLE_SIZE];
    uint32_t filter_cond_alias[VMK_FILTER_TABLE_SIZE * VMK_FILTER_TABLE_SIZE];
    float filter_cond_func[VMK_FILTER_TABLE_SIZE * VMK_FILTER_TABLE_SIZE];
    /* IlluminationIntegrator (integrator.cpp:59-66) */
    uint32_t max_depth, min_depth;
    float rr_threshold;
    uint32_t mis_mode; /* 0 both, 1 light, 2 bsdf (integrator.h:116-120) */
    /* LightSampler (lightsampler.cpp:11-14) */
    uint32_t env_separate;
    float env_prob;
    float ray_offset_factor; /* render_setting.ray_offset_factor (scene.cpp:33) */
    /* FrameBuffer (frame_buffer.cpp:15-26) */
    float exposure;
    uint32_t tone_mapper; /* 0 linear, 1 aces, 2 reinhard (tonemapper/impl.cpp:16-45) */
    /* participating media: Scene::process_mediums() (scene_desc.cpp:26-35) and the sensor's medium
     * (photosensory.cpp:11-32, sensor.cpp:48) */
    uint32_t process_mediums;
    uint32_t camera_medium; /* index into mediums[] or VMK_INVALID */
    uint32_t light_sampler; /* 0 "uniform" (uniform.cpp), 1 "power" (power.cpp; needs vmk_scene::light_alias_offset) */
} vmk_render_params;

/* Tile ownership for multi-GPU sharding: the image is cut into tile_size^2 tiles; tile (tx, ty) is rendered by this ctx
 * iff owner(tx, ty) == rank, with
 *     owner(tx, ty) = (tx + skew * ty) mod world,   skew = vmk_tile_skew(world)
 * (the smallest odd number >= 0.38 * world that is coprime to world: 1, 3, 5, 7 for world = 2, 4, 8, 16).  Every
 * world x world block of tiles is a Latin square: each rank owns exactly one tile per row and per column of the block, so
 * no rank is pinned to a column stripe (what `t mod world` gives when the tiles per row are a multiple of world, e.g. 120 at
 * 3840 px) or to a band of rows (what a bit-reversed tile index mod a power-of-two world degenerates to), whatever the
 * resolution.  DESIGN.md section 6. */
typedef struct vmk_tiles {
    uint32_t tile_size; /* pixels, power of two; 0 => whole image, single owner */
    uint32_t rank, world;
} vmk_tiles;
uint32_t vmk_tile_skew(uint32_t world);

typedef struct vmk_counters { /* cumulative since vmk_reset_counters */
    uint64_t closest_rays;    /* trace_closest calls */
    uint64_t shadow_rays;     /* trace_occlusion calls */
    uint64_t nodes_visited;   /* BVH node records fetched */
    uint64_t tris_tested;     /* triangle records fetched */
    uint64_t paths;           /* camera samples started */
    uint64_t surface_hits;    /* closest hits that were shaded */
    uint64_t tex_fetches;     /* bilinear texture lookups (4 texels each) */
} vmk_counters;

typedef struct vmk_ctx vmk_ctx;

/* ---- lifecycle ---------------------------------------------------------------------------------- */
int vmk_create(int device, vmk_ctx **out);
void vmk_destroy(vmk_ctx *ctx);
const char *vmk_last_error(const vmk_ctx *ctx); /* valid until the next call on ctx; ctx may be NULL */
uint32_t vmk_abi_version(void);

int vmk_upload_scene(vmk_ctx *ctx, const vmk_scene *scene);
int vmk_build_accel(vmk_ctx *ctx); /* GPU build: Morton codes -> radix sort -> PLOC merge rounds -> 128 B BVH4 nodes */
int vmk_set_render_params(vmk_ctx *ctx, const vmk_render_params *params);

/* Film. fb == NULL: the ctx owns a width*height float4 accumulation buffer. Otherwise fb is a DEVICE
 * pointer to width*height*4 floats owned by the caller (e.g. a torch tensor that is all-reduced). */
int vmk_set_framebuffer(vmk_ctx *ctx, void *fb_device);
int vmk_reset_accum(vmk_ctx *ctx);

/* Render frames [frame_begin, frame_begin+frame_count) of the owned tiles, fused with accumulation:
 * acc = lerp(1/(f+1), acc, L_f) per frame f in order (frame_buffer.cpp:117-126).  Asynchronous on the
 * ctx stream; kernel_ms (optional) receives the HIP-event time of the launch(es) and forces a sync.
 * Internally one work item is one path (pixel, frame): paths write radiance to per-frame staging planes
 * (frame_count x owned pixels x 16 B of device memory, at most 8 GiB per launch — larger batches are split)
 * and a resolve kernel folds the planes into the accumulation buffer in frame order. */
int vmk_render_batch(vmk_ctx *ctx, uint32_t frame_begin, uint32_t frame_count, const vmk_tiles *tiles,
                     float *kernel_ms);
int vmk_synchronize(vmk_ctx *ctx);
/* Launch timing without a host sync per batch: with timing enabled every vmk_render_batch(…, kernel_ms = NULL) brackets its
 * launches with a HIP-event pair on the ctx stream; vmk_collect_kernel_ms waits for the stream and returns the elapsed times
 * of the calls since the last collect (how bench.py measures the kernel while batches and exchanges overlap). */
int vmk_enable_kernel_timing(vmk_ctx *ctx, int enabled);
int vmk_collect_kernel_ms(vmk_ctx *ctx, float *out_ms, uint32_t max_count, uint32_t *count);
int vmk_download_accum(vmk_ctx *ctx, float *out_rgba /* width*height*4 */);
/* exposure -> tone map -> host RGBA float (final_picture 0: the output_buffer of pt.cpp:96-116);  final_picture 1: + the second tone map
 * and the sRGB curve of Pipeline::final_picture (pipeline.cpp:337-354, postprocessor.cpp:13-30);  2: + the second tone map only, as that
 * function does for output names ending in "exr" / "hdr" (vmk_host_final_picture_mode picks 1 or 2 from a file name) */
int vmk_tonemap(vmk_ctx *ctx, int final_picture, float *out_rgba);

int vmk_get_counters(vmk_ctx *ctx, vmk_counters *out);
int vmk_reset_counters(vmk_ctx *ctx);
/* nodes_visited / tris_tested are tallied in the traversal's innermost loops (2 VALU per step).  enabled = 0 selects the
 * megakernel instance without them for the following vmk_render_batch calls (those two counters then stay put; rays, paths,
 * hits and texture fetches are always counted).  Rays are a pure function of (pixel, frame), so a sibling launch over the
 * same frames with enabled = 1 returns the tallies of a timed launch exactly (bench.py does that).  Default: enabled. */
int vmk_set_traversal_counters(vmk_ctx *ctx, int enabled);
void *vmk_stream(vmk_ctx *ctx); /* hipStream_t the ctx launches on */

/* ---- multi-GPU: the path's one exchange step (the reference is single-GPU, no precedent) --------
 * One process (or thread) per GPU, one ctx each; image tiles are sharded (vmk_tiles) and every rank's framebuffer holds
 * its own tiles and zeros elsewhere, so the sum over ranks IS the image, bit for bit (x + 0 is exact).  RCCL is loaded
 * with dlopen on first use.  The collective runs on a ctx-owned second stream after everything rendered so far; the
 * render stream is free to start the next vmk_render_batch at once — only that batch's film resolve, the one writer of
 * the framebuffer, waits for the exchange — so the collective overlaps the next batch's megakernel. */
#define VMK_COMM_ID_BYTES 128
int vmk_comm_unique_id(void *id_out /* VMK_COMM_ID_BYTES; ncclGetUniqueId, rank 0 calls it and hands the bytes to the others */);
int vmk_comm_init(vmk_ctx *ctx, const void *unique_id, int rank, int world); /* ncclCommInitRank: the ctx owns the communicator */
int vmk_comm_adopt(vmk_ctx *ctx, void *nccl_comm); /* use the host application's own ncclComm_t (not destroyed by the ctx) */
/* recv_device: width*height*4 floats on this device, distinct from the framebuffer; receives the full image on every rank */
int vmk_allreduce_framebuffer(vmk_ctx *ctx, void *recv_device);            /* ncclAllReduce(SUM), 2(G-1)/G * S bytes per link */
int vmk_allgather_framebuffer(vmk_ctx *ctx, const vmk_tiles *tiles, void *recv_device); /* pack owned tiles -> ncclAllGather -> unpack, (G-1)/G * S */
int vmk_comm_synchronize(vmk_ctx *ctx); /* wait for the exchange stream */

/* ---- BVH introspection + traversal replay (SURVEY §8d traversal-only roofline) ------------------------ */
typedef struct vmk_accel_info {
    uint32_t n_nodes, n_leaves, node_bytes, tri_bytes;
    float build_ms;
    uint32_t depth;       /* exact worst-case number of pending entries on a ray's traversal stack */
    uint32_t stack_depth; /* entries of the per-ray LDS traversal stack; builds with depth > stack_depth are rejected */
} vmk_accel_info;
int vmk_accel_info_get(vmk_ctx *ctx, vmk_accel_info *out);

/* Trace n rays given as SoA device-resident copies of host arrays (origin xyz, dir xyz, tmax); writes
 * hits as {inst, prim, bary.x, bary.y} (inst == VMK_INVALID on miss).  any_hit != 0: occlusion query,
 * hit[i*4] = 1/0.  Returns kernel time of the traversal launch in *kernel_ms. */
int vmk_trace_rays(vmk_ctx *ctx, uint32_t n, const float *org_xyz, const float *dir_xyz, const float *tmax,
                   int any_hit, uint32_t *hit_out /* n*4 */, float *kernel_ms, uint32_t repeats);

/* ---- AOV pass (FrameBuffer::compile_compute_geom, src/base/sensor/frame_buffer.cpp:156-219) -----------------
 * Primary-hit planes of frame `frame` for denoisers / image tooling: shading normal (w = 1 on a hit, 0 on a
 * miss), MaterialEvaluator::albedo (material.cpp:91-98), emitted radiance (evaluate_hit_wi), each width*height
 * RGBA floats, linear depth = (world-to-camera * p).z (sensor.cpp:192-195), width*height floats, and the motion
 * vector p_film - prev_raster_coord(p) (frame_buffer.cpp:483-491, sensor.cpp:95-100; the previous camera is the
 * current one, so this is the lens / filter reprojection offset), width*height*2 floats.  Any output may be NULL.
 * spectrum/hero: albedo and emission are linear_srgb(spectrum, the pixel's wavelengths of this frame) (:169-170,192-203). */
int vmk_render_aov(vmk_ctx *ctx, uint32_t frame, float *normal_rgba, float *albedo_rgba, float *emission_rgba,
                   float *depth, float *motion_xy);

/* ---- albedo-table precompute (the reference's vision-precompute app, src/apps/precompute/main.cpp:24-41;
 * Material::precompute_lobe base/scattering/material.h:121-163; Lobe::integral_albedo lobe.cpp:13-33) --------
 * Integrates table `which` (0 PureReflection res^2, 1 Dielectric / 2 DielectricInv res^3 x 2 floats {total,
 * reflected}, 3 Specular res^3, 4 Coat res^3) with sample_num samples per texel on the GPU; out is a host
 * array in the layout vmk_luts expects.  Needs no scene.  tools/make_luts.py builds vision_amd/data/luts.bin
 * with it. */
int vmk_precompute_albedo(vmk_ctx *ctx, uint32_t which, uint32_t res, uint32_t sample_num, float *out);

/* ---- device-side unit entry points used by the parity tests (tests/ only) ------------------------------ */
/* Evaluate n independent work items of test `kind` on the GPU; in/out are plain float arrays with
 * in_stride/out_stride floats per item (layouts documented in tests/test_gpu_parity.py).
 * kind 7 is the ray capture that feeds vmk_trace_rays replays: in = {pixel x, pixel y, frame} as u32 bit
 * patterns, out = {vertex count, then 16 floats per path vertex: closest ray o.xyz d.xyz t_max 1 | shadow
 * ray o.xyz d.xyz t_max traced} for up to 24 vertices (out_stride >= 385). */
int vmk_test_eval(vmk_ctx *ctx, uint32_t kind, uint32_t n, const float *in, uint32_t in_stride, float *out,
                  uint32_t out_stride);

/* Toolchain self-check: renders frame 0 of up to max_pixels (0 = 4096) strided pixels with the megakernel variant the
 * scene selects and with the unit kernel (a separately compiled instance of the same path code) and compares the
 * radiance bit for bit.  Returns VMK_OK when they agree; the user's framebuffer and counters are left untouched. */
int vmk_self_check(vmk_ctx *ctx, uint32_t max_pixels, uint32_t *n_checked, uint32_t *n_mismatch);
/* By default the first vmk_render_batch after vmk_build_accel / vmk_set_render_params runs vmk_self_check(ctx, 256, ...) itself and
 * fails with its status if the two instances disagree (one 1-spp frame + 256 unit-kernel paths, once per scene): a host that never
 * heard of the toolchain's miscompiles is guarded anyway.  enabled = 0 opts out (benchmarks that time the very
 * first batch; hosts that call vmk_self_check themselves). */
int vmk_set_auto_self_check(vmk_ctx *ctx, int enabled);

#ifdef __cplusplus
}
#endif
#endif /* VMK_H */
