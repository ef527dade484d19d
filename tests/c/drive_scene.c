/* drive_scene.c — the whole hot path driven from plain C through include/vmk.h + include/vmk_host.h only (no Python, no C++):
 * load a Vision scene -> hand decoded images over (none needed here) -> upload -> GPU BVH build -> render params -> self check ->
 * render a batch -> download the linear film -> tone map -> save the picture (.png, .exr).  What a C / C++ host such as Vision's plugin stub does.
 * usage: drive_scene <scene.json> <width> <height> <frames> <out.f32>      (out: width*height*4 float32, the linear film)
 * exit status: 0 ok, 1 usage, 2 any vmk / vmk_host error (message on stderr). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vmk.h"
#include "vmk_host.h"

#define HOST_TRY(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, vmk_host_last_error()); return 2; } } while (0)
#define VMK_TRY(x) do { if ((x) != VMK_OK) { fprintf(stderr, "%s: %s\n", #x, vmk_last_error(ctx)); return 2; } } while (0)

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s scene.json width height frames out.f32 [lut_path]\n", argv[0]); return 1; }
    const uint32_t width = (uint32_t) atoi(argv[2]), height = (uint32_t) atoi(argv[3]), frames = (uint32_t) atoi(argv[4]);

    /* Importer::import_scene + Scene::prepare (host half): Vision JSON -> flat tables */
    vmk_host_options opt;
    memset(&opt, 0, sizeof opt);
    opt.width = width; opt.height = height; opt.max_depth = -1; opt.min_depth = -1; opt.procedural_env = 1;
    opt.lut_path = argc > 6 ? argv[6] : NULL;
    /* images the scene references: a host that has decoded them (Vision's ImagePool) registers the pixels before loading */
    char listed[4096];
    int n_listed = vmk_host_list_images(argv[1], listed, (uint32_t) sizeof listed);
    if (n_listed < 0) { fprintf(stderr, "vmk_host_list_images: %s\n", vmk_host_last_error()); return 2; }
    printf("images referenced: %s\n", n_listed ? listed : "(none)");
    vmk_host_scene *hs = NULL;
    HOST_TRY(vmk_host_load_scene(argv[1], &opt, &hs));
    const vmk_scene *tables = vmk_host_scene_tables(hs);
    const vmk_render_params *params = vmk_host_render_params(hs);
    printf("scene: %u triangles, %u instances, %u materials, %u lights; film %ux%u, max_depth %u\n", tables->n_tris, tables->n_instances,
           tables->n_materials, tables->n_lights, params->width, params->height, params->max_depth);

    vmk_ctx *ctx = NULL;
    if (vmk_create(0, &ctx) != VMK_OK) { fprintf(stderr, "vmk_create: %s\n", vmk_last_error(NULL)); return 2; }
    if (vmk_abi_version() != VMK_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
    VMK_TRY(vmk_upload_scene(ctx, tables));
    VMK_TRY(vmk_build_accel(ctx));
    vmk_accel_info info;
    VMK_TRY(vmk_accel_info_get(ctx, &info));
    printf("accel: %u BVH4 nodes, %u leaves, built in %.2f ms\n", info.n_nodes, info.n_leaves, info.build_ms);
    VMK_TRY(vmk_set_render_params(ctx, params));
    uint32_t checked = 0, bad = 0;
    VMK_TRY(vmk_self_check(ctx, 1024, &checked, &bad));
    VMK_TRY(vmk_reset_accum(ctx));
    VMK_TRY(vmk_reset_counters(ctx));
    float ms = 0.f;
    VMK_TRY(vmk_render_batch(ctx, 0, frames, NULL, &ms));
    vmk_counters c;
    VMK_TRY(vmk_get_counters(ctx, &c));
    printf("rendered %u frames in %.3f ms: %llu paths, %llu closest + %llu shadow rays; self check on %u pixels ok\n", frames, ms,
           (unsigned long long) c.paths, (unsigned long long) c.closest_rays, (unsigned long long) c.shadow_rays, checked);

    const size_t n = (size_t) params->width * params->height * 4;
    float *film = (float *) malloc(n * sizeof(float)), *picture = (float *) malloc(n * sizeof(float));
    if (!film || !picture) return 2;
    VMK_TRY(vmk_download_accum(ctx, film));
    VMK_TRY(vmk_tonemap(ctx, 1, picture)); /* Pipeline::final_picture */
    double mean = 0.0;
    for (size_t i = 0; i < n; i += 4) mean += picture[i] + picture[i + 1] + picture[i + 2];
    printf("final picture mean %.6f\n", mean / (double) (n / 4 * 3));
    /* Pipeline::save_result (pipeline.cpp:190-204): final_picture -> Image::save_image, for a .png and for an .exr name (no gamma) */
    char name[4096];
    snprintf(name, sizeof name, "%s.png", argv[5]);
    HOST_TRY(vmk_host_save_image(name, params->width, params->height, picture));
    snprintf(name, sizeof name, "%s.exr", argv[5]);
    VMK_TRY(vmk_tonemap(ctx, vmk_host_final_picture_mode(name), picture));
    HOST_TRY(vmk_host_save_image(name, params->width, params->height, picture));
    printf("saved %s.png and %s.exr\n", argv[5], argv[5]);
    FILE *f = fopen(argv[5], "wb");
    if (!f || fwrite(film, sizeof(float), n, f) != n) { fprintf(stderr, "cannot write %s\n", argv[5]); return 2; }
    fclose(f);
    free(film); free(picture);
    vmk_destroy(ctx);
    vmk_host_free_scene(hs);
    return 0;
}
