#!/usr/bin/env python3
"""Goldens from reference OUTPUT files (data only): pictures the reference itself rendered and ships, decoded to LINEAR radiance.

  res/render_scene/cbox/dispersion-hero.png                    -> tests/golden/cbox_prism_ref.npz
      = res/render_scene/cbox/cbox-prism.json exactly as shipped (glass sphere LASF9, checker back wall, spectrum/hero dimension 4,
      1024x1024; scenes/cbox/cbox-prism.json is that file).  Encoding found by fit on the diffuse walls: ONE ACES tone map
      (exposure 1) + sRGB; gallery/dispersion.png is the same picture through the double tone map of Pipeline::final_picture.
  res/render_scene/classroom/output-1024spp.png                -> tests/golden/classroom_ref_detail.npz
      band-passed luminance of the textured regions (notice boards, blackboard, lectern): texture ORIENTATION, not lighting (the scene's
      environment map is stripped from the checkout).
  res/render_scene/glass-of-water/glass-of-water-1024spp.png   -> tests/golden/glass_of_water_ref.npz
      Encoding: exposure 1 - exp(-x) + sRGB (not saved through final_picture).

  res/test_case/coffee/output.png                              -> tests/golden/coffee_ref.npz
      = res/test_case/coffee/vision_scene.json (scenes/coffee/; the glass carafe Mesh010.obj is not in the checkout) seen through the
      picture's camera (fov_y 20, look_at mirrored: scenes/coffee/vision_scene_refcam.json).  Encoding: sRGB only.

  gallery/staircase.png                                        -> tests/golden/staircase_ref.npz
      = scenes/staircase/vision_scene.json (no stand-in) through the picture's camera (fov_y 20, pitch mirrored, yaw fitted:
      scenes/staircase/vision_scene_refcam.json).  Encoding: sRGB only.

Stored per picture: `valid` = packed bits of the pixels whose 8-bit value is NOT saturated (any channel > 0.97, dilated by 8 / 4
pixels: an inverse tone map cannot recover clipped highlights — the image of the lamp in the sphere, the sparkles on the ice), and
`lin` = 8x8 block sums of the decoded linear radiance over the valid pixels (float32).  For the dispersion picture also `fringe`:
the high-passed red-blue chroma (R - B) / (R + G + B) inside the sphere, float16 — the colour fringes the hero spectrum produces at
the checker edges seen through the dispersive glass.

Pictures examined and NOT usable as pins (different scene state than any shipped file): res/render_scene/cbox/dispersion-srgb.png,
srgb.png, hero.png, hero2.png, dispersion-hero2.png (dark back wall / prism / box variants of the Cornell scene).
(Round 3, first half, also listed gallery/staircase.png here: "no crop or zoom of the shipped view reproduces it".  True — the picture
has another field of view and a mirrored pitch, see above.)   python tools/make_golden_refimage.py
"""
import os
import numpy as np
from PIL import Image
from scipy.ndimage import binary_dilation, gaussian_filter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/res/render_scene"
B = 8


def inv_srgb(y):
    return np.where(y <= 0.04045, y / 12.92, np.power((y + 0.055) / 1.055, 2.4))


def inv_aces(y):  # y = x (a x + b) / (x (c x + d) + e), tonemapper/impl.cpp:27-34
    a, b, c, d, e = 2.51, 0.03, 2.43, 0.59, 0.14
    y = np.clip(y, 0.0, 0.999)
    A, Bq, C = a - c * y, b - d * y, -e * y
    return (-Bq + np.sqrt(np.maximum(Bq * Bq - 4 * A * C, 0.0))) / (2 * A)


def inv_exposure(y):  # 1 - exp(-x * exposure), exposure 1 (frame_buffer.cpp:135-144)
    return -np.log(np.maximum(1.0 - y, 1e-5))


def block_sums(img, valid):
    h, w, _ = img.shape
    v = img * valid[..., None]
    return v[:h // B * B, :w // B * B].reshape(h // B, B, w // B, B, 3).sum((1, 3)).astype(np.float32)


def fringe_map(lin):
    """(R - B) / (R + G + B), lightly smoothed, minus its own low-pass: what is left are the colour fringes at edges."""
    img = gaussian_filter(lin, (1.5, 1.5, 0))
    c = (img[..., 0] - img[..., 2]) / (img.sum(2) + 1e-3)
    return c - gaussian_filter(c, 12)


if __name__ == "__main__":
    # ---- cbox-prism ----
    ref = np.asarray(Image.open(f"{REF}/cbox/dispersion-hero.png").convert("RGB")).astype(np.float64) / 255.0
    valid = ~binary_dilation(ref.max(2) > 0.97, iterations=8)
    lin = inv_aces(inv_srgb(ref))
    fr = fringe_map(lin)[330:650, 440:760]  # the sphere's bounding box (centre (600, 490), radius 208 px)
    out = os.path.join(ROOT, "tests", "golden", "cbox_prism_ref.npz")
    np.savez_compressed(out, valid=np.packbits(valid), lin=block_sums(lin, valid), fringe=fr.astype(np.float16), shape=np.array(ref.shape[:2]))
    print(out, "valid", valid.mean(), "mean", lin[valid].mean(0))
    # ---- classroom: texture detail ----
    # The environment map of this scene is stripped from the checkout, so its LIGHTING cannot be compared — but what the textures show can:
    # the two notice boards and the blackboard carry image textures on OBJ meshes (flip_uv, uv interpolation, JPEG decode, row order), and
    # their band-passed luminance (gaussian 1 px minus gaussian 6 px: local detail, indifferent to exposure and to smooth illumination) is
    # kept for the regions below.  A render whose texture lookups were mirrored or transposed decorrelates completely.
    ref = np.asarray(Image.open(f"{REF}/classroom/output-1024spp.png").convert("RGB")).astype(np.float64) / 255.0
    lum = ref @ np.array([0.2126, 0.7152, 0.0722])
    band = gaussian_filter(lum, 1.0) - gaussian_filter(lum, 6.0)
    regions = {"left_board": (296, 424, 244, 336), "right_board": (300, 420, 868, 948), "blackboard": (300, 420, 360, 860), "lectern": (390, 500, 352, 504)}
    out = os.path.join(ROOT, "tests", "golden", "classroom_ref_detail.npz")
    np.savez_compressed(out, shape=np.array(ref.shape[:2]), **{k: band[y0:y1, x0:x1].astype(np.float16) for k, (y0, y1, x0, x1) in regions.items()},
                        **{k + "_box": np.array(v) for k, v in regions.items()})
    print(out, {k: float(np.abs(band[y0:y1, x0:x1]).mean()) for k, (y0, y1, x0, x1) in regions.items()})
    # ---- coffee maker (res/test_case/coffee: the reference's own test case, expected picture next to the scene) ----
    # Encoding found by trying the candidates: plain sRGB of the linear accumulation, clipped at 1 (no exposure curve, no tone map) — with
    # it every region agrees in ABSOLUTE radiance, no scale fitted.  The camera of the picture is not the camera of the file as today's
    # reference reads it: the picture has fov_y 20 (the class default; the file says 25) and the look_at direction mirrored in yaw and
    # pitch — a render with exactly those two changes (scenes/coffee/vision_scene_refcam.json, written below) aligns with the picture
    # to the pixel (band-passed luminance, best correlation at zero offset), so it comes from an older reading of the same file.
    ref = np.asarray(Image.open("/root/reference/res/test_case/coffee/output.png").convert("RGB")).astype(np.float64) / 255.0
    valid = ~binary_dilation(ref.max(2) > 0.97, iterations=4)
    lin = inv_srgb(ref)
    out = os.path.join(ROOT, "tests", "golden", "coffee_ref.npz")
    np.savez_compressed(out, valid=np.packbits(valid), lin=block_sums(lin, valid), shape=np.array(ref.shape[:2]))
    print(out, "valid", valid.mean(), "mean", lin[valid].mean(0))
    import json, re
    src = os.path.join(ROOT, "scenes", "coffee", "vision_scene.json")
    d = json.loads(re.sub(r"//.*", "", open(src).read()))
    cp = d["camera"]["param"]
    pos, tgt = np.array(cp["transform"]["param"]["position"]), np.array(cp["transform"]["param"]["target_pos"])
    v = tgt - pos
    cp["transform"]["param"]["target_pos"] = (pos + np.array([-v[0], -v[1], v[2]])).tolist()
    cp["fov_y"] = 20.0
    json.dump(d, open(os.path.join(ROOT, "scenes", "coffee", "vision_scene_refcam.json"), "w"), indent=1)
    # ---- staircase (gallery/staircase.png; scenes/staircase/vision_scene.json loads with no stand-in) ----
    # The same older reading as the coffee picture — plain sRGB, fov_y 20 instead of the file's 35, pitch mirrored — but here the yaw of
    # the picture is neither the file's nor its mirror image: it was FITTED (one parameter, the x component of the viewing direction:
    # -0.0159 against the file's -0.0076; band-passed luminance correlates best at zero offset with it, 48 pixels off without).
    ref = np.asarray(Image.open("/root/reference/gallery/staircase.png").convert("RGB")).astype(np.float64) / 255.0
    valid = ~binary_dilation(ref.max(2) > 0.97, iterations=4)
    lin = inv_srgb(ref)
    out = os.path.join(ROOT, "tests", "golden", "staircase_ref.npz")
    np.savez_compressed(out, valid=np.packbits(valid), lin=block_sums(lin, valid), shape=np.array(ref.shape[:2]))
    print(out, "valid", valid.mean(), "mean", lin[valid].mean(0))
    src = os.path.join(ROOT, "scenes", "staircase", "vision_scene.json")
    d = json.loads(re.sub(r"//.*", "", open(src).read()))
    cp = d["camera"]["param"]
    pos, tgt = np.array(cp["transform"]["param"]["position"]), np.array(cp["transform"]["param"]["target_pos"])
    v = tgt - pos
    ang = np.radians(1.35)  # the fitted yaw, on top of the mirrored direction
    bx, bz = -v[0], v[2]
    cp["transform"]["param"]["target_pos"] = (pos + np.array([bx * np.cos(ang) + bz * np.sin(ang), -v[1], -bx * np.sin(ang) + bz * np.cos(ang)])).tolist()
    cp["fov_y"] = 20.0
    json.dump(d, open(os.path.join(ROOT, "scenes", "staircase", "vision_scene_refcam.json"), "w"), indent=1)
    # ---- glass-of-water ----
    ref = np.asarray(Image.open(f"{REF}/glass-of-water/glass-of-water-1024spp.png").convert("RGB")).astype(np.float64) / 255.0
    valid = ~binary_dilation(ref.max(2) > 0.97, iterations=4)
    lin = inv_exposure(inv_srgb(ref))
    out = os.path.join(ROOT, "tests", "golden", "glass_of_water_ref.npz")
    np.savez_compressed(out, valid=np.packbits(valid), lin=block_sums(lin, valid), shape=np.array(ref.shape[:2]))
    print(out, "valid", valid.mean(), "mean", lin[valid].mean(0))
