// drender.h — the megakernel: one path vertex of IlluminationIntegrator::Li (path_bounce) and the persistent-lane
// render kernel around it (k_render).  Included by vmk.hip (srgb spectrum) and, with VMK_HERO = 1 and the namespace
// renamed, by vmk_hero.hip (hero spectrum) — see dpath.h.
#pragma once
#include "dbvh.h"

namespace vmkd {

constexpr int kBlock = 256;
__device__ __forceinline__ uint32_t wave_count(bool p) { return (uint32_t) __popcll(__ballot(p)); } // lanes of the wave with p (uniform)
#ifndef VMK_WAVES_PER_SIMD
#define VMK_WAVES_PER_SIMD 6 // __launch_bounds__ 2nd argument of the megakernel (register budget = 512 / n per lane).  Measured on classroom
                             // (Mrays/s, 32-spp probe, round 2: by-value lobe routines, no SLP, no traversal tallies): 4: 2710, 5: 2990,
                             // 6: 3230, 7 / 8: 2390 (6.25 KB of LDS per wave caps a CU at 24 waves = 6 per SIMD, so 7 and 8 only lose registers).
                             // Round 1 (out-pointer lobe routines): 3: 2015, 4: 2206, 5: 2297, 6: 1825, 8: 1487.
#endif
#ifndef VMK_MEDIA_WAVES_PER_SIMD
#define VMK_MEDIA_WAVES_PER_SIMD VMK_WAVES_PER_SIMD
#endif

// ---------------------------------------------------------------------------------------------------------
// one path vertex: IlluminationIntegrator::Li loop body (base/integral/integrator.cpp:160-311), no media
// ---------------------------------------------------------------------------------------------------------
struct PathState {
    Ray ray;
    V3 L;     // linear sRGB
    Spec T;   // throughput: a spectrum over the path's wavelengths
    V3 prev_ng;
    float scatter_pdf, eta_scale;
    uint32_t bounces;
    uint32_t medium; // RayState::medium: the medium the current ray travels in (MEDIA variants only)
#if VMK_HERO
    Swl swl;         // RenderEnv::sampled_wavelengths (integrator.cpp:48-57); L is linear sRGB, T a spectrum over swl
#endif
};
__device__ __forceinline__ void path_begin(PathState &ps, const vmk_render_params *P) {
    ps.L = mk3(0.f); ps.T = mks(1.f); ps.scatter_pdf = 1e16f; ps.eta_scale = 1.f; ps.prev_ng = ps.ray.d; ps.bounces = 0;
    ps.medium = P->process_mediums ? P->camera_medium : VMK_INVALID; // sensor.cpp:48
}

// ---- homogeneous medium + Henyey-Greenstein (render_core/medium/homogeneous.cpp:30-70, interaction.h:136-139,
//      interaction.cpp:12-32,114-134, geometry.cpp:187-199) — §8f rank 1 ----
// sigma_t / sigma_s as spectra: decode_to_unbound_spectrum of the RGB coefficients (homogeneous.cpp:34,54-55)
VD Spec medium_sigma_t(const DScene &S, const vmk_medium *m SWL_P) { return spec_unbound(S, (ld3(m->sigma_a) + ld3(m->sigma_s)) * m->scale SWL_A); }
VD Spec medium_sigma_s(const DScene &S, const vmk_medium *m SWL_P) { return spec_unbound(S, ld3(m->sigma_s) * m->scale SWL_A); }
VD Spec exp3(Spec v) { return smap(v, [](float x) { return exp_(x); }); }
VD Spec medium_Tr(const DScene &S, const vmk_medium *m, float t SWL_P) { return exp3((-1.f * medium_sigma_t(S, m SWL_A)) * fmin_(RayTMax, t)); }
VD Spec geometry_Tr(const DScene &S, const vmk_render_params *P, const Ray &r, uint32_t medium SWL_P) {
    if (P->process_mediums && medium != VMK_INVALID) return medium_Tr(S, S.mediums + medium, length(r.d) * r.t_max SWL_A);
    return mks(1.f);
}
VD float phase_HG(float cos_theta, float g) {
    float denom = 1.f + sqr(g) + 2.f * g * cos_theta;
    return Inv4Pi * (1.f - sqr(g)) / (denom * sqrt_(denom));
}
VD V3 hg_sample(V3 wo, float g, Sampler &sampler, float *f_out) { // 2 draws
    V2 u = sampler.next_2d();
    float sqr_term = (1.f - sqr(g)) / (1.f + g - 2.f * g * u.x);
    float cos_theta = -(1.f + sqr(g) - sqr(sqr_term)) / (2.f * g);
    cos_theta = abs_(g) < 1e-3f ? 1.f - 2.f * u.x : cos_theta;
    float sin_theta = safe_sqrt(1.f - sqr(cos_theta));
    float phi = 2.f * Pi * u.y;
    V3 v1, v2;
    coordinate_system(wo, &v1, &v2);
    float sp, cp; sincos_(phi, &sp, &cp);
    V3 wi = sin_theta * cp * v1 + sin_theta * sp * v2 + cos_theta * wo;
    *f_out = phase_HG(cos_theta, g);
    return wi;
}
// One path vertex for every lane of the wave: ALL lanes call this convergently (the two traversals inside are
// wave-cooperative), `active` says whether the lane carries a live path.  Returns kPathEnd when the lane's path ends at
// this vertex, kPathGoOn when it continues, and kPathTail when the bounce loop has run to max_depth in a configuration
// with the supplement of integrator.cpp:302-307 (`only_direct && mis_mode == EBoth`, i.e. max_depth < 2): the caller
// compares the new ray's direction with the primary ray's (`primary_miss`, :178-183) and either ends the path or calls
// again; a call with ps.bounces >= max_depth runs that supplement — mis_bsdf(bounce, inner = false): trace, add the
// environment / emitter the BSDF sample reaches with its MIS weight, no next-event estimation, no further bounce.
// `dbg` (tests / ray capture only): 16 floats per vertex —
// [hit inst, prim, bary.xy | light pdf, bsdf pdf towards the light, sampled pdf, occluded | shadow ray o.xyz d.xyz t_max, traced].
enum : int { kPathGoOn = 0, kPathEnd = 1, kPathTail = 2 };
template<bool FULL, bool MEDIA, bool COUNT = true, bool DEEP = false>
__device__ __forceinline__ int path_bounce(const DScene &S, const vmk_render_params *P, WaveScratch *ws, PathState &ps, Sampler &sampler,
                                           DCounters &cnt, float *dbg, bool active) {
    const uint32_t max_depth = P->max_depth, min_depth = P->min_depth, mis_mode = P->mis_mode;
#if VMK_HERO
    Swl &swl = ps.swl;
#endif
    if (max_depth == 0) return kPathEnd; // `$for(&bounces, 0, max_depth)` never runs (uniform: P is a kernel argument)
    const bool tail = ps.bounces >= max_depth; // the supplement pass (see above)
    Hit hit;
    cnt.closest += wave_count(active); // (wave-uniform tallies: one scalar add for the wave, no per-lane register — see DCounters)
    bool found = traverse_wave<COUNT, DEEP, false>(S, ps.ray, active, ws, hit, cnt);
    if (dbg && active) { dbg[0] = u2f(hit.inst); dbg[1] = u2f(hit.prim); dbg[2] = hit.bary.x; dbg[3] = hit.bary.y; }
    bool shade = false; // the lane reached a surface with a material: NEE + scattering follow
    bool tally_hit = false;
    Interaction it;
    LightSample ls;
    Ray shadow_ray = {mk3(0.f), mk3(0.f, 0.f, 1.f), 0.f};
    if (active && !found) { // evaluate_miss integrator.cpp:137-158
        if (S.env_light != VMK_INVALID) {
            Spec tr = mks(1.f);
            if constexpr (MEDIA) { // :146-151: a ray that leaves the scene inside a medium is attenuated over world_diameter
                if (P->process_mediums) {
                    ps.ray.t_max = S.lights[S.env_light].world_diameter;
                    tr = geometry_Tr(S, P, ps.ray, ps.medium SWL_A);
                }
            }
            LightEval ev = light_evaluate_miss_wi(S, P, ps.ray.o, ps.ray.d, cnt SWL_A);
            float weight = MIS_weight(ps.scatter_pdf, ev.pdf);
            weight = mis_mode == 2 ? 1.f : (mis_mode == 1 ? (ps.bounces == 0 ? weight : 0.f) : weight);
            ps.L += spec_linear_srgb(S, (ev.L * tr * weight) * ps.T SWL_A);
        }
    }
    bool pass_through = false;
    bool has_phase = false; // MEDIA: the vertex is a scattering event inside the medium
    float phase_g = 0.f;
    uint32_t med_in = VMK_INVALID, med_out = VMK_INVALID; // MediumInterface of the vertex
    if (active && found) {
        compute_surface_interaction<true>(S, hit.tri, hit.inst, hit.prim, hit.bary, it);
        it.wo = normalize(-ps.ray.d);
        if constexpr (MEDIA) {
            med_in = S.instances[hit.inst].inside_medium; med_out = S.instances[hit.inst].outside_medium; // geometry.cpp:90
            ps.ray.t_max = length(it.pos - ps.ray.o) / length(ps.ray.d);                                   // geometry.h:64-69
            if (P->process_mediums && ps.medium != VMK_INVALID) { // HomogeneousMedium::sample, 2 draws (integrator.cpp:199-206)
                const vmk_medium *m = S.mediums + ps.medium;
                Spec sigma_t = medium_sigma_t(S, m SWL_A), sigma_s = medium_sigma_s(S, m SWL_A);
                uint32_t channel = (uint32_t) (sampler.next_1d() * (float) kSpecDim); if (channel > kSpecDim - 1u) channel = kSpecDim - 1u;
                float st_c = scomp(sigma_t, channel);
                float dist = -log_(1.f - sampler.next_1d()) / st_c;
                float t = fmin_(dist / length(ps.ray.d), ps.ray.t_max);
                bool sampled_medium = t < ps.ray.t_max;
                if (sampled_medium) { // Interaction(ray->at(t), -ray->direction(), true), init_phase, set_medium
                    it.pos = ps.ray.o + ps.ray.d * t; it.wo = -1.f * ps.ray.d; it.ng = mk3(0.f); it.uv = {0.f, 0.f};
                    it.mat_id = VMK_INVALID; it.light_id = VMK_INVALID; it.prim_id = VMK_INVALID; it.prim_area = 0.f;
                    has_phase = true; phase_g = m->g; med_in = ps.medium; med_out = ps.medium;
                }
                Spec tr = medium_Tr(S, m, t SWL_A);
                Spec density = sampled_medium ? sigma_t * tr : tr;
                float pdf = average(density);
                ps.T *= sampled_medium ? tr * sigma_s / pdf : tr / pdf;
            }
        }
        if (it.mat_id == VMK_INVALID && !has_phase) { // integrator.cpp:208-214: pass through, bounce not counted
            if constexpr (MEDIA) ps.medium = P->process_mediums ? (dot(it.ng, ps.ray.d) > 0.f ? med_out : med_in) : VMK_INVALID;
            ps.ray = spawn_ray(it.pos, it.ng, ps.ray.d);
            pass_through = true;
        } else {
            tally_hit = !has_phase && !tail;
            if (it.light_id != VMK_INVALID) { // integrator.cpp:221-231
                LightEval ev = light_evaluate_hit_wi(S, P, ps.ray.o, it, cnt SWL_A);
                float weight = MIS_weight(ps.scatter_pdf, ev.pdf);
                weight = mis_mode == 2 ? 1.f : (mis_mode == 1 ? (ps.bounces == 0 ? weight : 0.f) : weight);
                Spec tr = mks(1.f);
                if constexpr (MEDIA) tr = geometry_Tr(S, P, ps.ray, ps.medium SWL_A);
                ps.L += spec_linear_srgb(S, ev.L * ps.T * weight * tr SWL_A);
            }
            ps.prev_ng = it.ng;
            if (!tail) {
                // NEE (3 draws) + shadow ray
                ls = light_sample_wi(S, P, it.pos, sampler, cnt SWL_A);
                shadow_ray = spawn_ray_to(it.pos, it.ng, ls.p_light);
                shade = true;
            }
        }
    }
    cnt.hits += wave_count(tally_hit); cnt.shadow += wave_count(shade);
    Hit sh;
    bool occluded = traverse_wave<COUNT, DEEP, true>(S, shadow_ray, shade, ws, sh, cnt); // (the occlusion-query instance of the loop)
    if (!shade) return pass_through ? kPathGoOn : kPathEnd;
    Spec tr_shadow = mks(1.f);
    if constexpr (MEDIA) tr_shadow = geometry_Tr(S, P, shadow_ray, P->process_mediums ? (dot(it.ng, shadow_ray.d) > 0.f ? med_out : med_in) : VMK_INVALID SWL_A);
    V3 wi = normalize(ls.p_light - it.pos);
    ScatterEval se; BSDFSample bs;
    if (MEDIA && has_phase) { // integrator.cpp:271-279: the phase function stands in for the BSDF (2 draws)
        float f = phase_HG(dot(it.wo, wi), phase_g);
        se.f = mks(f); se.pdf = f; se.flags = 0;
        float fs;
        bs.wi = hg_sample(it.wo, phase_g, sampler, &fs);
        bs.eval.f = mks(fs); bs.eval.pdf = fs; bs.eval.flags = 0; bs.eta = 1.f;
    } else {
        // material: evaluate towards the light, then sample (direct_lighting integrator.cpp:20-37)
        MatCtx mc;
        mc.lobe_lds = lobe_lds_slot(ws);
        // (scenes with a "normal" slot anywhere select the FULL variants: the single-lobe kernels carry none of it — even as an
        // out-of-line call behind a flag test it cost the headline kernel 11 %)
        if constexpr (FULL) it.shading = compute_shading_frame(S, S.materials + it.mat_id, it, cnt);
        mat_prepare<FULL>(S, S.materials + it.mat_id, it, mc, cnt SWL_A);
#if VMK_HERO
        // SampledWavelengths::check_dispersive (spectrum.cpp:32-39, integrator.cpp:264): a dispersive lobe keeps the hero wavelength only
        if (S.materials[it.mat_id].type == VMK_MAT_GLASS && (S.materials[it.mat_id].flags & VMK_MATF_DISPERSIVE)) { Spec keep = mks(0.f); keep.x = swl.pdf.x; swl.pdf = keep; } // invalidation_secondary (spectrum.cpp:21-30)
#endif
        mat_evaluate_and_sample<FULL>(S, mc, it, wi, sampler, se, bs, cnt SWL_A);
    }
    if (dbg) {
        dbg[4] = ls.eval.pdf; dbg[5] = se.pdf; dbg[6] = bs.eval.pdf; dbg[7] = occluded ? 1.f : 0.f;
        dbg[8] = shadow_ray.o.x; dbg[9] = shadow_ray.o.y; dbg[10] = shadow_ray.o.z; dbg[11] = shadow_ray.d.x; dbg[12] = shadow_ray.d.y;
        dbg[13] = shadow_ray.d.z; dbg[14] = shadow_ray.t_max; dbg[15] = 1.f;
    }
    bool is_delta_light = ls.eval.pdf < 0.f;
    float weight = mis_mode != 1 ? (is_delta_light ? 1.f : MIS_weight(ls.eval.pdf, se.pdf)) : 1.f;
    ls.eval.pdf = is_delta_light ? -ls.eval.pdf : ls.eval.pdf;
    Spec Ld = mks(0.f);
    if (!occluded && se.pdf > 0.f && ls.eval.pdf > 0.f) Ld = ls.eval.L * se.f * weight / ls.eval.pdf;
    if (mis_mode == 2) Ld = Ld * 0.f;
    ps.L += spec_linear_srgb(S, ps.T * Ld * tr_shadow SWL_A);
    ps.eta_scale *= sqr(bs.eta);
    float lum = max_comp(ps.T);
    if (!(bs.eval.pdf > 0.f) || lum == 0.f) return kPathEnd;
    ps.T *= bs.eval.f / bs.eval.pdf;
    if (ps.eta_scale * lum < P->rr_threshold && ps.bounces >= min_depth) { // integrator.cpp:292-299
        float q = fmin_(0.95f, lum);
        float rr = sampler.next_1d();
        if (q < rr) return kPathEnd;
        ps.T = ps.T / q;
    }
    ps.scatter_pdf = bs.eval.pdf;
    if constexpr (MEDIA) ps.medium = P->process_mediums ? (dot(it.ng, bs.wi) > 0.f ? med_out : med_in) : VMK_INVALID; // interaction.cpp:114-123
    ps.ray = spawn_ray(it.pos, it.ng, bs.wi);
    ++ps.bounces;
    if (ps.bounces < max_depth) return kPathGoOn;
    return (max_depth < 2u && mis_mode == 0u) ? kPathTail : kPathEnd; // integrator.cpp:302 `only_direct && mis_mode_ == EBoth`
}
// `primary_miss` of the supplement pass (integrator.cpp:178-183): the pass is skipped when the ray it would trace still
// has the primary ray's direction.  The primary direction is not kept in registers for this one cold use: it is
// regenerated from the path's (pixel, frame) key, out of line (the call sits on a path only max_depth < 2 launches take).
__device__ __noinline__ bool tail_is_primary(const vmk_render_params *P, uint32_t px, uint32_t py, uint32_t frame, V3 d) {
    Sampler s; s.start(px, py, frame, 0);
    Ray r = generate_ray(P, px, py, s);
    return r.d.x == d.x && r.d.y == d.y && r.d.z == d.z;
}

// ---------------------------------------------------------------------------------------------------------
// the megakernel
// ---------------------------------------------------------------------------------------------------------
// Work decomposition of one render launch.  A work item is ONE path: item w -> frame frame_begin + w / n_slots of pixel
// slot w % n_slots (slot = owned tile k, Morton position inside the tile), so consecutive items are neighbouring pixels
// of the same frame.  Paths write their radiance to the staging plane stage[frame][slot]; k_film_resolve then folds the
// planes into the accumulation buffer in frame order (the film's running mean is order dependent).  Path-granular
// items keep every lane busy to the end of the launch however few pixels a GPU owns (1/8 of the image at 8 GPUs is
// fewer pixels than resident lanes) and whatever the cost spread between pixels.
// The scene view travels BY VALUE in the kernel arguments: its fields are then scalar loads from the kernarg segment, and
// the compiler knows its pointers are global-memory addresses (pointers inside a kernel-argument struct are coerced to the
// global address space; loaded from a DScene in memory they would be generic and every access a flat_load with 64-bit
// VALU address arithmetic, counted on lgkmcnt as well as vmcnt).  Measured on classroom: 2270 -> 2530 Mrays/s together with
// keeping the view out of scratch (see LobeLuts in dpath.h).
struct RenderRest {
    const vmk_render_params *params;
    float4 *accum;
    float4 *stage;        // [frame_count][n_slots]
    uint32_t *queue;      // work-item counter
    unsigned long long *counters; // 7 x u64 (vmk_counters layout)
    uint32_t frame_begin, frame_count;
    uint32_t tile_size, tile_shift, tiles_x, tiles_y, rank, world;
    uint32_t n_slots, n_items, chunk; // chunk: items a wave claims with one atomic
    const uint32_t *tile_table;       // the owned tiles (row-major tile indices, ascending) or null: every tile (world == 1)
#ifdef VMK_DIAG
    float *diag; // [n_items][8 vertices][16 floats]: the dbg record of path_bounce (diagnostic builds only)
#endif
};
struct RenderArgs : RenderRest { DScene scene; };

__device__ __forceinline__ uint32_t compact_bits(uint32_t v) { // inverse of 2-D Morton interleave (even bits)
    v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0F0F0F0Fu; v = (v | (v >> 4)) & 0x00FF00FFu; v = (v | (v >> 8)) & 0x0000FFFFu;
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ bool slot_to_pixel(const RenderRest &A, uint32_t slot, uint32_t width, uint32_t height, uint32_t *px, uint32_t *py) {
    uint32_t k = slot >> (2u * A.tile_shift), r = slot & (A.tile_size * A.tile_size - 1u);
    uint32_t tile = A.tile_table ? A.tile_table[k] : k;
    uint32_t tx = tile % A.tiles_x, ty = tile / A.tiles_x;
    *px = tx * A.tile_size + compact_bits(r); *py = ty * A.tile_size + compact_bits(r >> 1);
    return *px < width && *py < height;
}

// All four <FULL, MEDIA> variants run at the same waves/SIMD.  (Round 1 pinned the MEDIA variants to 4 after a
// k_render<true, true> at 96 registers disagreed with the unit kernel; the cause turned out to be the SLP-vectoriser
// miscompile described in DESIGN.md section 8, which the build now avoids with -fno-slp-vectorize.)
template<bool FULL, bool MEDIA, bool COUNT, bool DEEP>
__global__ __launch_bounds__(kBlock, MEDIA ? VMK_MEDIA_WAVES_PER_SIMD : VMK_WAVES_PER_SIMD) void k_render(RenderArgs A) {
    __shared__ WaveScratch s_ws[kBlock / 64];
    const DScene S = A.scene;
    const vmk_render_params *P = A.params;
    const uint32_t lane = threadIdx.x & 63u;
    WaveScratch *ws = s_ws + (threadIdx.x >> 6);
    DCounters cnt = {0, 0, 0, 0, 0, 0, 0};

    // wave-uniform: the part of the wave's claimed chunk not yet handed to a lane
    uint32_t w_lo = 0, w_hi = 0;
    bool exhausted = false;
    // per-lane persistent state
    bool has_path = false;
    uint32_t item = 0;
    Sampler sampler; sampler.state = 0;
    PathState ps;
    ps.ray = {mk3(0.f), mk3(0.f, 0.f, 1.f), 0.f};
    ps.L = mk3(0.f); ps.T = mks(1.f); ps.prev_ng = mk3(0.f); ps.scatter_pdf = 1e16f; ps.eta_scale = 1.f; ps.bounces = 0; ps.medium = VMK_INVALID;
#if VMK_HERO
    ps.swl.lambda = mks(538.f); ps.swl.pdf = mks(1.f);
#endif

#ifdef VMK_DIAG
    uint32_t diag_gen = 0, diag_verts = 0; // paths this lane has finished / vertices of the current one, returned in the film's alpha
#endif
    for (;;) {
        // ---- hand new paths to idle lanes: ballot + prefix inside the wave, one atomic per chunk ----
        const unsigned long long need_mask = __ballot(!has_path);
        if (need_mask && (w_lo < w_hi || !exhausted)) {
            if (w_lo >= w_hi) {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(A.queue, A.chunk);
                b = (uint32_t) __builtin_amdgcn_readfirstlane((int) b);
                w_lo = min(b, A.n_items); w_hi = (uint32_t) min((unsigned long long) b + A.chunk, (unsigned long long) A.n_items);
                if ((unsigned long long) b + A.chunk >= A.n_items) exhausted = true;
            }
            bool started = false;
            if (!has_path) {
                uint32_t w = w_lo + (uint32_t) __popcll(need_mask & ((1ull << lane) - 1ull));
                if (w < w_hi) {
                    uint32_t f = w / A.n_slots, slot = w - f * A.n_slots, px, py;
                    if (slot_to_pixel(A, slot, P->width, P->height, &px, &py)) {
                        // ray generation (rt_geom kernel of the reference, frame_buffer.cpp:172-177)
                        uint32_t frame = A.frame_begin + f;
                        sampler.start(px, py, frame, 0);
                        ps.ray = generate_ray(P, px, py, sampler);
#if VMK_HERO
                        sampler.start(px, py, frame, 0xFFFFFFFFu); // RenderEnv::initial: start(pixel, frame, -1) inside temporary()
                        ps.swl = sample_wavelengths(sampler);
#endif
                        sampler.start(px, py, frame, 1); // path_tracing kernel, integrator.cpp:93
                        path_begin(ps, P);
                        has_path = true; item = w; started = true;
                    }
                }
            }
            cnt.paths += wave_count(started);
            w_lo = min(w_lo + (uint32_t) __popcll(need_mask), w_hi);
        }
        if (!__any(has_path) && exhausted && w_lo >= w_hi) break;

        // ---- one bounce of IlluminationIntegrator::Li (integrator.cpp:160-311) ----
        // (all lanes take part: the traversals inside are wave-cooperative; lanes without a path contribute no ray)
#ifdef VMK_DIAG
        float dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int state = path_bounce<FULL, MEDIA, COUNT, DEEP>(S, P, ws, ps, sampler, cnt, dbg, has_path);
        if (has_path && diag_verts < 8u && A.diag) {
            float *q = A.diag + ((size_t) item * 8u + diag_verts) * 16u;
            for (int k = 0; k < 16; ++k) q[k] = dbg[k];
            q[8] = ps.ray.o.x; q[9] = ps.ray.o.y; q[10] = ps.ray.o.z; q[11] = ps.ray.d.x; q[12] = ps.ray.d.y; q[13] = ps.ray.d.z; q[14] = ps.T.x;
            q[15] = u2f(sampler.state);
        }
#else
        int state = path_bounce<FULL, MEDIA, COUNT, DEEP>(S, P, ws, ps, sampler, cnt, nullptr, has_path);
#endif
        if (state == kPathTail && has_path) { // max_depth < 2 only (uniform per launch)
            uint32_t f = item / A.n_slots, px, py;
            (void) slot_to_pixel(A, item - f * A.n_slots, P->width, P->height, &px, &py);
            state = tail_is_primary(P, px, py, A.frame_begin + f, ps.ray.d) ? kPathEnd : kPathGoOn;
        }
        const bool terminate = state != kPathGoOn;
#ifdef VMK_DIAG
        if (has_path) ++diag_verts;
#endif
        if (has_path && terminate) {
#ifdef VMK_DIAG
            A.stage[item] = make_float4(ps.L.x, ps.L.y, ps.L.z, (float) (diag_gen * 100u + diag_verts));
            ++diag_gen; diag_verts = 0;
#else
            A.stage[item] = make_float4(ps.L.x, ps.L.y, ps.L.z, 1.f);
#endif
            has_path = false;
        }
    }
    // ---- counters: one atomic per wave and counter ----
    // (closest, shadow, paths and hits are wave totals already; the traversal tallies and the texture fetches are per lane)
    uint32_t c[7] = {cnt.closest, cnt.shadow, wave_sum(cnt.nodes), wave_sum(cnt.tris), cnt.paths, cnt.hits, wave_sum(cnt.tex)};
#pragma unroll
    for (int i = 0; i < 7; ++i) if (lane == 0 && c[i]) atomicAdd(A.counters + i, (unsigned long long) c[i]);
}

// ---------------------------------------------------------------------------------------------------------
// unit kernel: the whole path of one (pixel, frame) stepped vertex by vertex with path_bounce<true, true>, one lane per path
// and no persistent loop — a second, independently compiled instance of the path code (its own register budget), which
// vmk_self_check compares bit for bit with the megakernel variant a scene selects, and which the parity tests compare with
// the oracle's path records.  Output: 8 floats per vertex for the first 8 vertices, then L (o[64..66]).
// ---------------------------------------------------------------------------------------------------------
constexpr int kUnitPathVertexCap = 1 << 16; // a safety net far above anything a real path reaches (k_render has no cap either)
// Between two vertices the unit kernel keeps its path state in LDS, not in registers: every iteration reloads (ray, L, T, ...)
// through volatile accesses and stores them back after path_bounce.  The twin then shares no loop-carried register allocation with
// the megakernel it checks — the place both miscompiles of this toolchain showed up — and costs the
// megakernel nothing.
constexpr uint32_t kUnitStateDwords = 22u + 3u * kSpecDim; // ray 7, L 3, prev_ng 3, T, 4 scalars, sampler, (hero: lambda, pdf), padding for srgb
struct UnitState { uint32_t w[kUnitStateDwords][64]; };
__device__ __forceinline__ void unit_state_store(UnitState *us, const PathState &ps, const Sampler &smp) {
    volatile uint32_t *q = &us->w[0][threadIdx.x & 63u];
    uint32_t k = 0;
    auto put = [&](float v) { q[(k++) * 64u] = f2u(v); };
    put(ps.ray.o.x); put(ps.ray.o.y); put(ps.ray.o.z); put(ps.ray.d.x); put(ps.ray.d.y); put(ps.ray.d.z); put(ps.ray.t_max);
    put(ps.L.x); put(ps.L.y); put(ps.L.z); put(ps.prev_ng.x); put(ps.prev_ng.y); put(ps.prev_ng.z);
    put(ps.scatter_pdf); put(ps.eta_scale); put(u2f(ps.bounces)); put(u2f(ps.medium)); put(u2f(smp.state));
#pragma unroll
    for (uint32_t i = 0; i < kSpecDim; ++i) put(scomp(ps.T, i));
#if VMK_HERO
#pragma unroll
    for (uint32_t i = 0; i < kSpecDim; ++i) { put(scomp(ps.swl.lambda, i)); put(scomp(ps.swl.pdf, i)); }
#endif
}
__device__ __forceinline__ void unit_state_load(const UnitState *us, PathState &ps, Sampler &smp) {
    const volatile uint32_t *q = &us->w[0][threadIdx.x & 63u];
    uint32_t k = 0;
    auto get = [&]() { return u2f(q[(k++) * 64u]); };
    ps.ray.o.x = get(); ps.ray.o.y = get(); ps.ray.o.z = get(); ps.ray.d.x = get(); ps.ray.d.y = get(); ps.ray.d.z = get(); ps.ray.t_max = get();
    ps.L.x = get(); ps.L.y = get(); ps.L.z = get(); ps.prev_ng.x = get(); ps.prev_ng.y = get(); ps.prev_ng.z = get();
    ps.scatter_pdf = get(); ps.eta_scale = get(); ps.bounces = f2u(get()); ps.medium = f2u(get()); smp.state = f2u(get());
#pragma unroll
    for (uint32_t i = 0; i < kSpecDim; ++i) sput(ps.T, i, get());
#if VMK_HERO
#pragma unroll
    for (uint32_t i = 0; i < kSpecDim; ++i) { sput(ps.swl.lambda, i, get()); sput(ps.swl.pdf, i, get()); }
#endif
}
__device__ __forceinline__ void unit_path(const DScene &S, const vmk_render_params *P, WaveScratch *ws, UnitState *us, bool live, uint32_t px, uint32_t py, uint32_t frame, float *o, DCounters &cnt) {
    {
        Sampler smp; smp.start(px, py, frame, 0);
        PathState ps; ps.ray = generate_ray(P, px, py, smp);
#if VMK_HERO
        smp.start(px, py, frame, 0xFFFFFFFFu); // RenderEnv::initial
        ps.swl = sample_wavelengths(smp);
#endif
        smp.start(px, py, frame, 1);
        path_begin(ps, P);
        unit_state_store(us, ps, smp);
    }
    bool alive = live;
    for (int v = 0; v < kUnitPathVertexCap && __any(alive); ++v) { // wave-uniform trip count: path_bounce is wave-cooperative
        float dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        PathState ps; Sampler smp;
        unit_state_load(us, ps, smp);
        int st = path_bounce<true, true, true, true>(S, P, ws, ps, smp, cnt, dbg, alive); // (DEEP: the unit kernel serves every tree)
        if (alive) unit_state_store(us, ps, smp);
        if (alive && v < 8) for (int k = 0; k < 8; ++k) o[v * 8 + k] = dbg[k];
        if (st == kPathTail && alive) st = tail_is_primary(P, px, py, frame, ps.ray.d) ? kPathEnd : kPathGoOn;
        if (st != kPathGoOn) alive = false;
    }
    if (live) {
        PathState ps; Sampler smp;
        unit_state_load(us, ps, smp);
        o[64] = ps.L.x; o[65] = ps.L.y; o[66] = ps.L.z;
    }
}
// in: 3 uint32 (px, py, frame) per path; out: >= 67 floats per path; launched with 64-thread blocks
__global__ void k_unit_path(const DScene *scene, const vmk_render_params *P, uint32_t n, const float *in, uint32_t in_stride, float *out, uint32_t out_stride) {
    __shared__ WaveScratch s_ws[1];
    __shared__ UnitState s_us[1];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const float *a = in + (size_t) (live ? i : 0) * in_stride;
    float *o = out + (size_t) (live ? i : 0) * out_stride;
    DCounters cnt = {0, 0, 0, 0, 0, 0, 0};
    const DScene S = *scene;
    unit_path(S, P, s_ws, s_us, live, f2u(a[0]), f2u(a[1]), f2u(a[2]), o, cnt);
}

// ---------------------------------------------------------------------------------------------------------
// AOVs of the primary hit — FrameBuffer::compile_compute_geom, the `rt_geom` kernel (frame_buffer.cpp:156-219): shading normal,
// linear depth, motion vector, MaterialEvaluator::albedo and the emitted radiance, the last two brought to linear sRGB through the
// pixel's wavelengths when the spectrum is hero.  One thread per pixel; the primary rays go through the same wave-cooperative
// traversal as the megakernel's.
// ---------------------------------------------------------------------------------------------------------
struct AovArgs {
    const DScene *scene;
    const vmk_render_params *params;
    float4 *normal, *albedo, *emission; // RGBA planes or null
    float *depth;
    float2 *motion;
    uint32_t frame;
    float w2s[16]; // inverse(c2w)              (Sensor::store_prev_data sensor.cpp:89-93; the camera is static within a render)
    float s2r[16]; // inverse(raster_to_sensor)
};
__global__ __launch_bounds__(kBlock) void k_aov(AovArgs A) {
    __shared__ WaveScratch s_ws[kBlock / 64];
    const DScene S = *A.scene;
    const vmk_render_params *P = A.params;
    WaveScratch *ws = s_ws + (threadIdx.x >> 6);
    DCounters cnt = {0, 0, 0, 0, 0, 0, 0};
    const uint32_t n = P->width * P->height;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += gridDim.x * blockDim.x) { // block-uniform trip count
        const uint32_t i = base + threadIdx.x;
        const bool live = i < n;
        const uint32_t px = live ? i % P->width : 0u, py = live ? i / P->width : 0u;
#if VMK_HERO
        Sampler wl; wl.start(px, py, A.frame, 0xFFFFFFFFu); // RenderEnv::initial (frame_buffer.cpp:169-170): the pixel's wavelengths for this frame
        const Swl swl = sample_wavelengths(wl);
#endif
        Sampler sampler; sampler.start(px, py, A.frame, 0);
        V2 p_film;
        Ray ray = generate_ray(P, px, py, sampler, &p_film);
        Hit hit;
        bool found = traverse_wave<true, true, false>(S, ray, live, ws, hit, cnt); // (DEEP: serves every tree)
        if (!live) continue;
        V3 normal = mk3(0.f), albedo = mk3(0.f), emission = mk3(0.f);
        float depth = 0.f;
        V2 motion = {0.f, 0.f};
        if (found) {
            Interaction it;
            compute_surface_interaction<true>(S, hit.tri, hit.inst, hit.prim, hit.bary, it);
            it.wo = normalize(-ray.d);
            normal = it.shading.z;
            depth = A.w2s[2] * it.pos.x + A.w2s[6] * it.pos.y + A.w2s[10] * it.pos.z + A.w2s[14];
            { // compute_motion_vec (frame_buffer.cpp:483-491) against Sensor::prev_raster_coord (sensor.cpp:95-100)
                V3 ps = transform_point4(A.w2s, it.pos);
                ps = ps / ps.z;
                V3 rc = transform_point4(A.s2r, ps);
                motion = {p_film.x - rc.x, p_film.y - rc.y};
            }
            if (it.mat_id != VMK_INVALID) {
                MatCtx mc; // (the albedo of a lobe does not depend on its shading frame: no compute_shading_frame here)
                mat_prepare<true>(S, S.materials + it.mat_id, it, mc, cnt SWL_A);
                albedo = spec_linear_srgb(S, mat_albedo(S, mc, it, cnt SWL_A) SWL_A); // frame_buffer.cpp:192-196
            }
            if (it.light_id != VMK_INVALID) emission = spec_linear_srgb(S, light_evaluate_hit_wi(S, P, ray.o, it, cnt SWL_A).L SWL_A); // :197-203
        }
        if (A.normal) A.normal[i] = make_float4(normal.x, normal.y, normal.z, found ? 1.f : 0.f);
        if (A.albedo) A.albedo[i] = make_float4(albedo.x, albedo.y, albedo.z, 1.f);
        if (A.emission) A.emission[i] = make_float4(emission.x, emission.y, emission.z, 1.f);
        if (A.depth) A.depth[i] = depth;
        if (A.motion) A.motion[i] = make_float2(motion.x, motion.y);
    }
}


// the sixteen ahead-of-time variants: <FULL, MEDIA> x {tallying, not tallying} x {trees that fit the LDS stack, deep trees with the HBM
// stack overflow}
typedef void (*RenderKernel)(RenderArgs);
template<bool COUNT, bool DEEP>
inline RenderKernel select_render_kernel_cd(bool full, bool media) {
    return full ? (media ? k_render<true, true, COUNT, DEEP> : k_render<true, false, COUNT, DEEP>) : (media ? k_render<false, true, COUNT, DEEP> : k_render<false, false, COUNT, DEEP>);
}
inline RenderKernel select_render_kernel(bool full, bool media, bool count, bool deep) {
    if (deep) return count ? select_render_kernel_cd<true, true>(full, media) : select_render_kernel_cd<false, true>(full, media);
    return count ? select_render_kernel_cd<true, false>(full, media) : select_render_kernel_cd<false, false>(full, media);
}

}// namespace vmkd
