"""Audio readers for the two container formats the reference accepts (GammatoneFiltering.py:28-39):
RIFF/WAVE (via scipy.io.wavfile, as the reference does) and NIST SPHERE, which is what TIMIT's ".WAV"
files really are (the reference goes through the `sphfile` package and a per-sample Python loop)."""
import numpy
from scipy.io import wavfile as _wavfile


def read_sphere(filename):
    """(sample_rate, int16 samples) from an uncompressed 16-bit PCM NIST SPHERE file (first channel)."""
    with open(filename, "rb") as f:
        head = f.read(16)
        if not head.startswith(b"NIST_1A"):
            raise ValueError(f"{filename}: neither a RIFF nor a NIST SPHERE file")
        header_size = int(head.split(b"\n")[1].strip())
        f.seek(0)
        text = f.read(header_size).decode("ascii", errors="replace")
        fields = {}
        for line in text.split("\n")[2:]:
            parts = line.strip().split(None, 2)
            if not parts or parts[0] == "end_head":
                break
            if len(parts) == 3:
                fields[parts[0]] = parts[2]
        coding = fields.get("sample_coding", "pcm")
        if coding != "pcm":
            raise ValueError(f"{filename}: SPHERE sample_coding '{coding}' is not supported (only uncompressed pcm)")
        if int(fields.get("sample_n_bytes", 2)) != 2:
            raise ValueError(f"{filename}: only 16-bit SPHERE files are supported")
        order = "<i2" if fields.get("sample_byte_format", "01") == "01" else ">i2"
        channels = int(fields.get("channel_count", 1))
        count = int(fields["sample_count"]) if "sample_count" in fields else -1
        data = numpy.fromfile(f, dtype=order, count=count * channels if count >= 0 else -1)
    if channels > 1:
        data = data.reshape(-1, channels)[:, 0]
    return int(fields.get("sample_rate", 16000)), numpy.ascontiguousarray(data, dtype=numpy.int16)


def write_sphere(filename, rate, samples):
    """Minimal SPHERE writer (tests and synthetic TIMIT-like corpora)."""
    samples = numpy.ascontiguousarray(samples, dtype="<i2")
    body = ("NIST_1A\n   1024\nchannel_count -i 1\nsample_count -i %d\nsample_rate -i %d\nsample_n_bytes -i 2\n"
            "sample_byte_format -s2 01\nsample_sig_bits -i 16\nsample_coding -s3 pcm\nend_head\n"
            % (samples.shape[0], rate)).encode("ascii")
    with open(filename, "wb") as f:
        f.write(body.ljust(1024, b" "))
        samples.tofile(f)


def read_audio(filename):
    with open(filename, "rb") as f:
        magic = f.read(4)
    if magic == b"RIFF":
        return _wavfile.read(filename)
    return read_sphere(filename)
