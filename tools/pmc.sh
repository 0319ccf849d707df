#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> <counter list> -- <python args...>
# one rocprofv3 --pmc pass (counters only; kernel-trace is a separate run), CSV output
out=$1; shift; ctrs=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/$out -- python3 "$@" > gpurun_out/$out.log 2>&1
