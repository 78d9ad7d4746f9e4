#!/bin/bash
# copy the summaries of one tools/profile_round4.sh run from gpurun_out/prof_<tag>/ into profiles/ under the names profiles/README.md uses
#   bash tools/collect_profiles.sh r04c        (C3)        bash tools/collect_profiles.sh r04c_c5   (C5)
tag=$1; src=gpurun_out/prof_$tag; [ -d "$src" ] || { echo "no $src"; exit 1; }
cp $src/roofline_plain.json profiles/${tag}_roofline_only.json
cp $src/roofline_traced.json profiles/${tag}_roofline_only_under_rocprofv3.json
cp $src/bench_default.json profiles/${tag}_bench_default.json
cp $(find $src/roofline -name "*kernel_stats.csv" | head -1) profiles/${tag}_roofline_only_kernel_stats.csv
cp $(find $src/default -name "*kernel_stats.csv" | head -1) profiles/${tag}_bench_default_kernel_stats.csv
cp $src/counters_per_kernel.json profiles/${tag}_counters_per_kernel.json
cp $src/k_binary_traffic.json profiles/${tag}_k_binary_traffic.json
case $tag in *_c5) cp $src/k_binary_traffic.json profiles/k_binary_traffic_c5.json ;; *) cp $src/k_binary_traffic.json profiles/k_binary_traffic_c3.json; cp $src/k_binary_traffic.json profiles/k_binary_traffic.json ;; esac
ls -la profiles/${tag}_*
