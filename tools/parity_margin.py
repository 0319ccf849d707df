"""Print the measured parity margin of the float-FFT envelope (per-channel max-norm error vs the oracle) for a few
row lengths, with and without the low-pass. Diagnostic only: the asserted bound lives in tests/."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
from conftest import chan_relerr
from f2cnn_amd import _lib
from f2cnn_amd.scripts.processing import EnvelopeExtraction as EE
from oracle import f2cnn_oracle as orc

for n in (1500, 3000, 8000, 16000, 16384, 30000, 40000, 70001, 140000):
    wav = orc.synth_utterance(n, n)
    gfb = orc.erb_filterbank(wav, orc.make_erb_filters(16000, orc.centre_freqs(16000, 128, 100)))
    for lpf in (False, True):
        ref = orc.extract_envelope_from_matrix(gfb, lpf, 50)
        e32 = chan_relerr(EE.ExtractEnvelopeFromMatrix(gfb, lpf, 50, precision=_lib.FFT_F32), ref)
        e64 = chan_relerr(EE.ExtractEnvelopeFromMatrix(gfb, lpf, 50, precision=_lib.FFT_F64), ref)
        print(f"n={n:6d} lpf={int(lpf)}  float FFT {e32:.2e}   double FFT {e64:.2e}", flush=True)
