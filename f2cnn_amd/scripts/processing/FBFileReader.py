"""Reader for the VTR formant database's .FB files (reference scripts/processing/FBFileReader.py): a 12-byte
big-endian HTK-like header (nFrames int32, sampPeriod int32, sampSize int16, fileType int16) followed by 8
big-endian float32 per frame (F1..F4, B1..B4 in kHz). Host-side; SURVEY section 8f row n3."""
import numpy


def round_half_even_decimal(values, digits):
    """Python's round(x, digits) for every element of a float64 array.

    The reference rounds each value with the built-in round() on a Python float (:44-47), which rounds the exact
    decimal expansion of the binary value; numpy.round computes rint(x * 10^digits) / 10^digits, and the product can
    land on the other side of a tie. The two agree unless x * 10^digits is within rounding distance of k + 0.5, so
    only those elements (none in a typical file) take the per-element path."""
    values = numpy.asarray(values, dtype=numpy.float64)
    out = numpy.round(values, digits)
    scaled = numpy.abs(values) * 10.0 ** digits
    near_tie = numpy.abs(scaled - numpy.floor(scaled) - 0.5) < 1e-6
    if near_tie.any():
        flat, src = out.reshape(-1), values.reshape(-1)
        for i in numpy.flatnonzero(near_tie.reshape(-1)):
            flat[i] = round(float(src[i]), digits)
    return out


def ExtractFBFile(fbFilename, verbose=False):
    """(nFrame, 8) float64 matrix in Hz rounded to 2 decimals, and the sampling period (fixed 10000 us like the
    reference, which ignores the header field because one VTR file carries a wrong value, :22-24)."""
    try:
        with open(fbFilename, 'rb') as fbFile:
            header = fbFile.read(12)
            nFrame = int(numpy.frombuffer(header[0:4], dtype='>i4')[0])
            sampSize = int(numpy.frombuffer(header[8:10], dtype='>i2')[0])
            fileType = int(numpy.frombuffer(header[10:12], dtype='>i2')[0])
            sampPeriod = 10000
            if verbose:
                print('N_SAMPLES=', nFrame)
                print('SAMP_PERIOD=', sampPeriod)
                print('SAMP_SIZE=', sampSize)
                print('NUM_COMPS=', sampSize / 4)
                print('FILE_TYPE=', fileType)
            data = numpy.fromfile(fbFile, dtype='>f4', count=nFrame * 8)
        if data.size != nFrame * 8:
            raise ValueError("{}: truncated .FB file ({} of {} values)".format(fbFilename, data.size, nFrame * 8))
        # kHz float32 -> float64 -> Hz, rounded to 2 decimals the way the built-in round() does
        return round_half_even_decimal(data.astype(numpy.float64).reshape(nFrame, 8) * 1000, 2), sampPeriod
    except FileNotFoundError:
        print("No .FB formant data file.")
        return None, 0


def GetFormantFrequencies(fbFilename, formant):
    """Track of formant 1..4 in Hz (one value per frame) and the sampling period; (None, None) without a file."""
    matrix, sampPeriod = ExtractFBFile(fbFilename)
    return (None, None) if matrix is None else (matrix[:, formant - 1], sampPeriod)


def frame_windows(track, timepoints, radius, samples_per_frame):
    """Rows of 2*radius+1 consecutive frames of `track`, one per WAV sample index in `timepoints`: the window of
    timepoint t starts at frame int(t / samples_per_frame - radius) (reference :69-89). A window that would start
    before frame 0, or whose end int(t / samples_per_frame + radius) + 1 reaches len(track), ends the program as the
    reference does."""
    track = numpy.asarray(track)
    t = numpy.asarray(timepoints, dtype=numpy.float64) / samples_per_frame
    first = numpy.trunc(t - radius).astype(numpy.int64)
    last = numpy.trunc(t + radius).astype(numpy.int64) + 1
    width = 2 * radius + 1
    bad = (first < 0) | (last >= len(track))
    if bad.any():
        i = int(numpy.flatnonzero(bad)[0])
        print("ERROR: formant window [{}, {}) of timepoint {} (radius {}) is outside the {} frames of the track".format(
            int(first[i]), int(last[i]), int(numpy.asarray(timepoints).reshape(-1)[i]), radius, len(track)))
        exit(-1)
    if len(first) == 0 or len(track) < width:
        return numpy.zeros((len(first), width), dtype=track.dtype)
    return numpy.lib.stride_tricks.sliding_window_view(track, width)[first]


def GetFromantFrequenciesAround(array, timepoint, radius, wavToFormant):
    """The 2*radius+1 frame values centred on WAV sample `timepoint` (the reference's name, typo included)."""
    start = int(timepoint / wavToFormant - radius)
    end = int(timepoint / wavToFormant + radius) + 1
    frame_windows(array, [timepoint], radius, wavToFormant)          # range check (exits like the reference)
    return array[start:end]
