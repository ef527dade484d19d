// vmk_hero.hip — the megakernel compiled for the hero-wavelength spectrum (render_core/spectrum/hero.cpp).
//
// Vision JIT-compiles its path-tracing kernel against the scene's Spectrum plugin; here the two spectra are two
// ahead-of-time instances of the same source (drender.h and the headers under it): this translation unit sets
// VMK_HERO = 1 and renames the device namespace, so that the sRGB instance in vmk.hip — the one the headline benchmark
// runs — keeps exactly the code, register budget and symbols it has without this file.  libvmk.so links both.
//
// The spectrum's dimension (HeroWavelengthSpectrum::dimension_, hero.cpp:240) is a build parameter as well: this file is the
// 3-wavelength instance; vmk_hero4.hip defines VMK_SPEC_DIM = 4 and includes it again for `"dimension": 4` scenes
// (cbox-prism.json:692-697), with its own namespace and entry-point prefix.
#define VMK_HERO 1
#ifndef VMK_SPEC_DIM
#define VMK_SPEC_DIM 3
#endif
#if VMK_SPEC_DIM == 4
#define vmkd vmkd_hero4
#define HERO_FN(name) vmk_hero4_##name
#else
#define vmkd vmkd_hero
#define HERO_FN(name) vmk_hero_##name
#endif
#include "drender.h"

#include <cstring>

hipError_t HERO_FN(occupancy)(bool full, bool media, bool count, bool deep, int *blocks_per_cu) {
    using namespace vmkd;
    auto kernel = select_render_kernel(full, media, count, deep);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kernel, kBlock, 0);
}
// Launch k_render<FULL, MEDIA> of the hero instance.  `rest` points at vmk.hip's RenderRest, `scene` at its DSceneFull (same
// declarations, same layouts: only the namespace differs).
hipError_t HERO_FN(launch_render)(bool full, bool media, bool count, bool deep, unsigned blocks, hipStream_t stream, const void *rest, size_t rest_bytes, const void *scene, size_t scene_bytes) {
    using namespace vmkd;
    RenderArgs A;
    if (rest_bytes != sizeof(RenderRest) || scene_bytes != sizeof(DScene)) return hipErrorInvalidValue;
    std::memcpy(static_cast<RenderRest *>(&A), rest, sizeof(RenderRest));
    std::memcpy(&A.scene, scene, sizeof(DScene));
    auto kernel = select_render_kernel(full, media, count, deep);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBlock), 0, stream, A);
    return hipGetLastError();
}
// the hero instance of the path unit kernel (drender.h k_unit_path): vmk_test_eval kind 6 and vmk_self_check on hero scenes
hipError_t HERO_FN(launch_unit_path)(hipStream_t stream, const void *scene, const void *params, uint32_t n, const float *in, uint32_t in_stride, float *out, uint32_t out_stride) {
    using namespace vmkd;
    hipLaunchKernelGGL(k_unit_path, dim3((n + 63) / 64), dim3(64), 0, stream, (const DScene *) scene, (const vmk_render_params *) params, n, in, in_stride, out, out_stride);
    return hipGetLastError();
}
// the hero instance of the AOV pass (drender.h k_aov): albedo and emission come back as linear sRGB through the pixel's wavelengths
hipError_t HERO_FN(launch_aov)(unsigned blocks, hipStream_t stream, const void *args, size_t args_bytes) {
    using namespace vmkd;
    AovArgs A;
    if (args_bytes != sizeof(AovArgs)) return hipErrorInvalidValue;
    std::memcpy(&A, args, sizeof(AovArgs));
    hipLaunchKernelGGL(k_aov, dim3(blocks), dim3(kBlock), 0, stream, A);
    return hipGetLastError();
}
