// Third translation unit of K2: f2_envelope.hip compiled with its kernel body as the device function envelope_row, for
// k_envelope_flagged - the launch that serves utterances the spectral kernel's accuracy guard sends back (normally none)
// with 1 / 16 of the workgroups. In its own namespace (as f2_envelope_p3.hip) so that the main translation unit's
// kernels, which serve everything the spectral kernel does not, compile exactly as before; exports only
// f2_launch_envelope_flagged (f2_internal.h).
#define F2_ENVELOPE_FLAGGED_TU
#define f2fft f2fft_fl
#include "f2_envelope.hip"
