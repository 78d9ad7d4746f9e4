# dev tool: kernel-trace timeline of the default pipelined bench + two schedule variants on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { tag=$1; shift
out=gpurun_out/ovl_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/d -- python3 bench.py --steps 30 --warmup 3 --cpu-frames 0 --no-extras "$@" > $out/b.json 2> $out/err
echo "=== $tag: $@ (RMCV_PIXEL_GROUPS=$RMCV_PIXEL_GROUPS RMCV_SPARSE_WAVES=$RMCV_SPARSE_WAVES)"
tail -1 $out/b.json | cut -c60-130
python3 tools/ovl_timeline.py $out/d
rm -rf $out/d
}
one default
one ctx6 --streams 6 --sparse-streams 3
RMCV_PIXEL_GROUPS=3 one g3
