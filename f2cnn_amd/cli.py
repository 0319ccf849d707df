"""Command line of the hot path, same surface as the reference's f2cnn.py (:68-160) for the commands this
package implements:

    python -m f2cnn_amd prepare filter
    python -m f2cnn_amd prepare envelope [--cutoff/-c HZ]
    python -m f2cnn_amd prepare label                       (needs the VTR .FB and TIMIT .PHN side files)
    python -m f2cnn_amd prepare input [--cutoff HZ] [--label/-l CSV] [--input/-i NPY]
    python -m f2cnn_amd prepare features [--cutoff HZ]     (filter + envelope in one pass, not in the reference)
    (filter / envelope / features also take --skip-existing to resume and --metrics FILE for a JSON summary; a file that
     cannot be read is reported and skipped, the exit status is then 2)
    python -m f2cnn_amd cnn train [--input/-i NPY] [--label/-l CSV]   (PyTorch-ROCm autograd; weights -> last_trained_model)
    python -m f2cnn_amd cnn eval --file/-f WAV [--lpf HZ] [--model/-m NPZ]
    python -m f2cnn_amd cnn evalnoise --file/-f WAV --noise/-n SNRdB [--lpf HZ] [--model/-m NPZ]
    python -m f2cnn_amd cnn evalrand [--count/-c N] [--lpf HZ] [--model/-m NPZ]
    python -m f2cnn_amd --configure            (writes configF2CNN.conf with the reference's defaults)

organize / plot need the licensed TIMIT+VTR corpora or matplotlib and stay with the reference.
"""
import argparse

PREPARE = ("filter", "envelope", "label", "input", "features")
CNN = ("train", "eval", "evalnoise", "evalrand")


def build_parser():
    parser = argparse.ArgumentParser(prog="f2cnn_amd", description="F2CNN hot path on MI355X.")
    parser.add_argument('--configure', action='store_true', help='write configF2CNN.conf with default values')
    sub = parser.add_subparsers()
    p = sub.add_parser('prepare', help='data processing commands')
    p.add_argument('--cutoff', '-c', action='store', dest='CUTOFF', type=int,
                   help="low pass filter the envelopes with this cutoff frequency")
    p.add_argument('prepare_command', choices=PREPARE)
    p.add_argument('--file', '-f', dest='file', nargs='?')
    p.add_argument('--input', '-i', dest='inputFile', nargs='?')
    p.add_argument('--label', '-l', dest='labelFile', nargs='?')
    p.add_argument('--skip-existing', action='store_true',
                   help="filter / envelope / features: leave files whose outputs are already up to date (resume)")
    p.add_argument('--metrics', dest='metrics', nargs='?',
                   help="filter / envelope / features: write files, audio seconds, wall time and audio-s/s as JSON")
    c = sub.add_parser('cnn', help='CNN commands')
    c.add_argument('--file', '-f', dest='file', nargs='?')
    c.add_argument('--input', '-i', dest='inputFile', nargs='?')
    c.add_argument('--label', '-l', dest='labelFile', nargs='?')
    c.add_argument('--model', '-m', dest='model', nargs='?')
    c.add_argument('cnn_command', choices=CNN)
    c.add_argument('--lpf', action='store', type=int, dest='CUTOFF', help="low pass filter the envelopes")
    c.add_argument('--count', '-c', action='store', type=int, help="number of files to evaluate (evalrand)")
    c.add_argument('--noise', '-n', action='store', type=float, dest='SNRdB', help="SNR in dB (evalnoise)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if 'prepare_command' in args:
        kwargs = {}
        if args.prepare_command in ('envelope', 'input', 'features'):      # f2cnn.py:112-114
            kwargs['LPF'] = args.CUTOFF is not None
            kwargs['CUTOFF'] = args.CUTOFF
        if args.prepare_command == 'input':
            if args.labelFile is not None:
                kwargs['labelFile'] = args.labelFile
            if args.inputFile is not None:
                kwargs['inputFile'] = args.inputFile
        if args.prepare_command in ('filter', 'envelope', 'features'):
            kwargs['skip_existing'] = args.skip_existing
            kwargs['metrics'] = args.metrics
        if args.prepare_command == 'filter':
            from .scripts.processing.GammatoneFiltering import FilterAllOrganisedFiles as fn
        elif args.prepare_command == 'envelope':
            from .scripts.processing.EnvelopeExtraction import ExtractAllEnvelopes as fn
        elif args.prepare_command == 'label':
            from .scripts.processing.LabelDataGenerator import GenerateLabelData as fn
        elif args.prepare_command == 'features':
            from .scripts.processing.EnvelopeExtraction import FilterAndExtractAll as fn
        else:
            from .scripts.processing.InputGenerator import GenerateInputData as fn
        report = fn(**kwargs)
        if getattr(report, "exit_status", 0):          # some files could not be processed (the others were)
            return report.exit_status
    elif 'cnn_command' in args:
        if args.cnn_command == 'train':                        # f2cnn.py:126-143
            import os
            from .scripts.CNN.Training import TrainAndPlotLoss
            inputFile = args.inputFile or args.file or os.path.join('trainingData', 'last_input_data.npy')
            labelFile = args.labelFile or os.path.join('trainingData', 'label_data.csv')
            if not os.path.isfile(inputFile):
                print("Please first generate the input data file with 'prepare input', or give one with --input")
                return 1
            if not os.path.isfile(labelFile):
                print("Please first generate a label data file with 'prepare label', or give one with --label")
                return 1
            TrainAndPlotLoss(labelFile=labelFile, inputFile=inputFile)
            return 0
        from .scripts.CNN import Evaluating
        kwargs = {}
        if args.CUTOFF is not None:                            # f2cnn.py:149-151
            kwargs['LPF'] = True
            kwargs['CUTOFF'] = args.CUTOFF
        if args.model is not None:
            kwargs['model'] = args.model
        if args.cnn_command == 'evalrand':                     # needs no --file (unreachable in the reference CLI)
            if args.count is not None:
                kwargs['count'] = args.count
            Evaluating.EvaluateRandom(**kwargs)
            return 0
        if args.file is None:
            print("Please use --file or -f to give input file")
            return 1
        kwargs['file'] = args.file
        if args.cnn_command == 'evalnoise':
            if args.SNRdB is not None:
                kwargs['SNRdB'] = args.SNRdB
            Evaluating.EvaluateWithNoise(**kwargs)
        else:
            Evaluating.EvaluateOneWavFile(**kwargs)
    elif args.configure:
        from .config import write_default
        print("Saving configuration file as '{}'".format(write_default()))
    else:
        print("No valid command given.")
        print("For help, use python -m f2cnn_amd --help")
        return 1
    return 0
