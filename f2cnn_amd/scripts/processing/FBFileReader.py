"""Reader for the VTR formant database's .FB files (reference scripts/processing/FBFileReader.py): a 12-byte
big-endian HTK-like header (nFrames int32, sampPeriod int32, sampSize int16, fileType int16) followed by 8
big-endian float32 per frame (F1..F4, B1..B4 in kHz). Host-side; SURVEY section 8f row n3."""
import numpy


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
        # float32 -> Python float -> *1000 -> round(.., 2), element by element in the reference (:44-47)
        return numpy.round(data.astype(numpy.float64).reshape(nFrame, 8) * 1000, 2), sampPeriod
    except FileNotFoundError:
        print("No .FB formant data file.")
        return None, 0


def GetFormantFrequencies(fbFilename, formant):
    """Frequencies (Hz) of formant 1..4 for every frame, and the sampling period."""
    matrix, sampPeriod = ExtractFBFile(fbFilename)
    if matrix is not None:
        return matrix[:, formant - 1], sampPeriod
    return None, None


def GetFromantFrequenciesAround(array, timepoint, radius, wavToFormant):
    """The 2*radius+1 frame values centred on WAV sample `timepoint` (reference :69-89; same name, typo included)."""
    start, end = timepoint / wavToFormant - radius, timepoint / wavToFormant + radius
    start, end = int(start), int(end) + 1
    if start < 0 or end >= len(array):
        print("ERROR: WRONG RANGE IN GETFORMANTFREQUENCIESAROUND IN ARRAY OF LEN:\n", len(array), "\nAT TIME AND RADIUS",
              timepoint, radius, "START", start, "END", end)
        exit(-1)
    return array[start:end]
