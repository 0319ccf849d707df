"""Reader for TIMIT .PHN phoneme segmentation files (reference scripts/processing/PHNFileReader.py): one
"<first sample> <last sample> <phoneme>" line per segment. Host-side; SURVEY section 8f row n3.

The names below are the reference's module interface (`LabelDataGenerator` skips the SILENTS; the plotting code of the
reference colours by class): phoneme classes of the TIMIT documentation as word lists."""

_CLASSES = {
    "STOPS": "b d g p t k dx q",
    "AFFRICATIVES": "jh ch",
    "FRICATIVES": "s sh w wh f th v dh",
    "NASALS": "m n ng em en eng nx",
    "SEMIVOWELS_AND_GLIDES": "l r w y hh hv el",
    "VOWELS": "iy ih eh ey ae aa aw ay ah ao oy ow uh uw ux er ax ix axr ax-h",
    "SILENTS": "pau epi h#",
}
globals().update({name: words.split() for name, words in _CLASSES.items()})
SILENTS = _CLASSES["SILENTS"].split()        # (spelled out for readers and linters; same list as above)


def ExtractPhonemes(phnFilename):
    """[(phoneme, first sample, last sample), ...] in file order, or None (with a message) when the file is missing."""
    try:
        with open(phnFilename, 'r') as phnFile:
            rows = (line.rstrip('\n').split(' ') for line in phnFile)
            return [(r[2], int(r[0]), int(r[1])) for r in rows if len(r) >= 3]
    except FileNotFoundError:
        print("No .PHN phoneme data file.")
        return None


def GetPhonemeFromArrayAt(phonemes, timepoint):
    """Phoneme whose segment contains sample `timepoint` (both ends inclusive; the first such segment in file order
    wins where segments touch), 'h#' outside every segment."""
    return next((name for name, first, last in phonemes if first <= timepoint <= last), 'h#')


def GetPhonemeAt(phnFilename, timepoint):
    return GetPhonemeFromArrayAt(ExtractPhonemes(phnFilename), timepoint)
