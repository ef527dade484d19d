/*
 * vmk_host.h — C-ABI of the C++ host that keeps Vision's scene front-end (JSON schema, plugin type names and
 * per-type defaults) and encodes a scene into the flat tables of include/vmk.h instead of emitting DSL.
 *
 * Reference interfaces replaced (Vision `src/`, file:line):
 *   vmk_host_load_scene      Importer::import_scene -> SceneDesc::from_json -> Scene::init / Scene::prepare /
 *                            Pipeline::prepare_geometry (host half)      importers/json/importer.cpp:16-23,
 *                                                                         base/import/scene_desc.cpp:37-62,
 *                                                                         base/import/node_desc.cpp (defaults),
 *                                                                         base/mgr/scene.cpp:16-35,79-91,165-187
 *   vmk_host_register_image  ocarina Image::load (image_pool.cpp:23-28)   decoded pixels handed in by the caller take precedence;
 *                                                                         otherwise .png (8-bit), baseline .jpg, .hdr and .exr are decoded
 *                                                                         natively (csrc/host/image_codec.h, csrc/host/exr.h)
 *   vmk_host_load_image / vmk_host_save_image / vmk_host_final_picture_mode
 *                            ocarina Image::load / Image::save_image; Pipeline::save_result + final_picture
 *                                                                         base/mgr/pipeline.cpp:190-204,337-354
 * Error convention: 0 ok, negative on error, message via vmk_host_last_error() (thread-local).
 */
#ifndef VMK_HOST_H
#define VMK_HOST_H

#include "vmk.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vmk_host_scene vmk_host_scene;

typedef struct vmk_host_options {
    uint32_t width, height;     /* 0 = keep pipeline.param.frame_buffer.param.resolution */
    int32_t max_depth;          /* <0 = keep integrator.param.max_depth */
    int32_t min_depth;          /* <0 = keep */
    uint32_t procedural_env;    /* 1: a missing environment image is replaced by the seeded procedural sky */
    uint32_t drop_unsupported_lights; /* 1: skip light types outside the hot-path scope (point/spot/projector) instead of failing */
    const char *lut_path;       /* albedo-table blob (vision_amd/data/luts.bin); NULL = default next to the library */
    uint32_t mediums;           /* 0: ignore the scene's "mediums" block (the non-fog variant, BASELINE config 3);
                                 * 1: honour it (mediums.process, global medium, per-shape inside/outside, sensor medium) */
    uint32_t spectrum;          /* 0: keep the scene's "spectrum" block; 1: force spectrum/srgb; 2: force spectrum/hero (the shipped
                                 * scenes carry the hero line commented out — this flips it without editing the file), with the block's
                                 * "dimension" (3 when absent); 3: force spectrum/hero with "dimension": 4 */
    uint32_t missing_assets;    /* 0: a mesh / texture file the scene names but the disk lacks is an error; 1 ("standin"): a missing mesh is
                                 * skipped and a missing texture becomes a 1x1 mid-grey constant, each listed by vmk_host_describe as
                                 * "... (stand-in: file missing ...)" — for scenes whose large assets are stripped from the reference
                                 * checkout (bathroom2: 10 meshes, WoodPanel.png, the HDRI) */
} vmk_host_options;

/* Register decoded pixels for an image file so the loader does not need a decoder for it.  `path` is matched
 * against the absolute path the scene resolves (scene_dir / fn).  8-bit: channels interleaved, is_float = 0;
 * float: is_float = 1.  Pixels are copied. */
int vmk_host_register_image(const char *path, uint32_t width, uint32_t height, uint32_t channels, int is_float,
                            const void *pixels);
void vmk_host_clear_images(void);

/* List the image files a scene references (absolute paths, '\n' separated) so a caller can decode + register
 * them.  Returns the number of bytes written (excluding the terminator) or a negative status. */
int vmk_host_list_images(const char *json_path, char *buf, uint32_t buf_bytes);

int vmk_host_load_scene(const char *json_path, const vmk_host_options *opt, vmk_host_scene **out);
void vmk_host_free_scene(vmk_host_scene *scene);

const vmk_scene *vmk_host_scene_tables(const vmk_host_scene *scene);
const vmk_render_params *vmk_host_render_params(const vmk_host_scene *scene);
uint32_t vmk_host_output_spp(const vmk_host_scene *scene);    /* output.spp (node_desc.cpp:360-369) */
const char *vmk_host_output_fn(const vmk_host_scene *scene);  /* output.fn */
/* one line per plugin object the scene instantiated: "category/type name" (Vision's plugin namespace) */
const char *vmk_host_describe(const vmk_host_scene *scene);

/* Image files without a scene — ocarina's Image::load / Image::save_image behind src/base/mgr/image_pool.cpp:23-28 and
 * Pipeline::save_result (src/base/mgr/pipeline.cpp:190-198).  Containers: .png (8-bit, non-interlaced), baseline .jpg, Radiance .hdr and
 * OpenEXR .exr (single-part scanline; NONE / RLE / ZIPS / ZIP / PIZ; half, float, uint; csrc/host/exr.h) are decoded; .png (8-bit RGB,
 * value * 255 + 0.5), .exr (float32 B G R, ZIP) and .hdr (RGBE) are written.  Anything else is an error with a message, never a guess.
 * vmk_host_load_image: interleaved pixels (`channels` of uint8 when *is_float == 0, of float32 otherwise), allocated by the library —
 * release with vmk_host_free_image. */
int vmk_host_load_image(const char *path, uint32_t *width, uint32_t *height, uint32_t *channels, int *is_float, void **pixels);
void vmk_host_free_image(void *pixels);
/* The save path of Pipeline::save_result without Python:  mode = vmk_host_final_picture_mode(fn)  (1: second tone map + sRGB gamma,
 * 2: second tone map only — names ending in "exr" / "hdr", pipeline.cpp:337-340);  vmk_tonemap(ctx, mode, rgba);
 * vmk_host_save_image(fn, width, height, rgba)  (4 floats per pixel, alpha ignored). */
int vmk_host_final_picture_mode(const char *fn);
int vmk_host_save_image(const char *path, uint32_t width, uint32_t height, const float *rgba);

/* Regenerate the sRGB -> sigmoid-spectrum coefficient table the hero spectrum uplifts colours with
 * (sRGBToSpectrumTable_Data, hero.cpp:52-76; its header "srgb2spec.h" is not part of the reference checkout) from the CIE
 * tables in `spectra_path` (vision_amd/data/spectra.bin) and write it to `out_path` (vision_amd/data/srgb2spec.bin, which
 * vmk_host_load_scene reads for spectrum/hero scenes).  threads = 0: all cores.  See csrc/host/rgb2spec_opt.h. */
int vmk_host_build_rgb2spec(const char *spectra_path, const char *out_path, uint32_t threads);

const char *vmk_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
