// Second translation unit of K2: f2_envelope.hip compiled with the three-pass plan 16-32-16 for H = 8192 complex points
// (the 1 s / 16 kHz row), exporting only f2_launch_envelope13_p3 (declared in f2_internal.h). The radix plans are
// compile-time functions of F2_PLAN13_PASSES, so the second plan gets its own copy of every template - in its own
// namespace (the rename below), otherwise the two translation units would define the same inline functions differently
// and the linker would keep one of each. Which plan a call gets is decided in f2_launch_envelope (f2_envelope.hip).
#define F2_PLAN13_PASSES 3
#define F2_ENVELOPE_P3_TU
#define f2fft f2fft_p3
#include "f2_envelope.hip"
