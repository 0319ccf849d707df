"""Diagnostic: bench.py against another build of the library (tools/build_variant.sh):
python tools/bench_with_lib.py LIB [bench args]."""
import os
import sys

if __name__ == "__main__":      # (bench.py starts worker processes that import the main module again)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from f2cnn_amd import build
    build.LIB_PATH = os.path.abspath(sys.argv[1])
    sys.argv = ["bench.py"] + sys.argv[2:]
    import bench
    bench.main()
