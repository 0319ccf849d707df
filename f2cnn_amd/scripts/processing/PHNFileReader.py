"""Reader for TIMIT .PHN phoneme segmentation files (reference scripts/processing/PHNFileReader.py): one
"start end phoneme" line per segment, sample indices. Host-side; SURVEY section 8f row n3."""

STOPS = ['b', 'd', 'g', 'p', 't', 'k', 'dx', 'q']
AFFRICATIVES = ['jh', 'ch']
FRICATIVES = ['s', 'sh', 'w', 'wh', 'f', 'th', 'v', 'dh']
NASALS = ['m', 'n', 'ng', 'em', 'en', 'eng', 'nx']
SEMIVOWELS_AND_GLIDES = ['l', 'r', 'w', 'y', 'hh', 'hv', 'el']
VOWELS = ["iy", "ih", "eh", "ey", "ae", "aa", "aw", "ay", "ah", "ao",
          "oy", "ow", "uh", "uw", "ux", "er", "ax", "ix", "axr", "ax-h"]
SILENTS = ['pau', 'epi', 'h#']


def ExtractPhonemes(phnFilename):
    """List of (phoneme, start, end) tuples, or None when the file is missing."""
    try:
        with open(phnFilename, 'r') as phnFile:
            data = []
            for line in phnFile:
                parts = line.rstrip('\n').split(' ')
                if len(parts) >= 3:
                    data.append((parts[2], int(parts[0]), int(parts[1])))
        return data
    except FileNotFoundError:
        print("No .PHN phoneme data file.")
        return None


def GetPhonemeFromArrayAt(phonemes, timepoint):
    """First segment containing `timepoint` (both ends inclusive), 'h#' when none does."""
    for phoneme, start, end in phonemes:
        if start <= timepoint <= end:
            return phoneme
    return 'h#'


def GetPhonemeAt(phnFilename, timepoint):
    return GetPhonemeFromArrayAt(ExtractPhonemes(phnFilename), timepoint)
