#!/bin/bash
# A/B of library builds inside one GPU call: tools/ab.sh "<bench args>" name1 name2 ...  ("main" = the in-tree library)
args=$1; shift
for n in "$@"; do
  echo -n "$n: "
  if [ "$n" = main ]; then bash tools/bench_kernels.sh $args; else RND_LIBRARY=$PWD/build_ab/lib_$n.so bash tools/bench_kernels.sh $args; fi
done
