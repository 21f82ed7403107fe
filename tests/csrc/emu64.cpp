// Double-precision build of the engine emulation (emu.cpp): the SAME source - wave FFT, engine,
// element-wise stages, table builders - with every `float` compiled as `double`.  TEST
// INFRASTRUCTURE.  It answers one question: is the engine's iteration the algorithm of
// librosa.griffinlim, or only close to it?  In double precision the two agree to ~1e-10 after
// 60 iterations, so what separates the fp32 engine from the fp32 oracle at large iteration
// counts is rounding, amplified by the momentum iteration, and nothing else.
//
// The system headers are included first, with the real `float`; the macros below then apply to
// the project's headers only (their own includes of the same system headers are guarded).
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <utility>
#include <vector>

#define float double
#define fmaf fma
#define sqrtf sqrt
#define fabsf fabs
#define fmaxf fmax
#define cosf cos
#define sinf sin
#include "emu.cpp"
