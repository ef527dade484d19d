// vmk_hero4.hip — the hero-spectrum megakernel with FOUR wavelengths per path (spectrum/hero, "dimension": 4:
// cbox-prism.json:692-697, cbox_debug.json:1113; HeroWavelengthSpectrum::sample_wavelength hero.cpp:286-299 rotates the hero
// wavelength by i / dimension).  Same source as vmk_hero.hip, compiled with Spec = four floats (dmath.h).
#define VMK_SPEC_DIM 4
#include "vmk_hero.hip"
