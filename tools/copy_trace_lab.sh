# Where does a needle_hip_library_stream_pcm call spend its time?  rocprofv3 memory-copy + kernel trace of
# tools/library_stream_device.py, then per call the large H2D copies: durations, gaps, where they get slower.
# usage: tools/copy_trace_lab.sh <episodes> <batch> <minutes>
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ct
NEEDLE_HIP_TRACE=1 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d /tmp/ct -- python3 $GRAFT_REPO_ROOT/tools/library_stream_device.py ${1:-750} ${2:-250} ${3:-45} > /tmp/ct.log 2>&1
grep "stream_pcm:\|episodes\":" /tmp/ct.log | cut -c1-200
python3 $GRAFT_REPO_ROOT/tools/copy_trace_analyze.py ${2:-250}
