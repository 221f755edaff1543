#!/bin/bash
# Same-box A/B of the working tree's library against the library of a git revision (default HEAD):
#   bash profiles/ab.sh [rev] -- builds <rev>'s csrc into csrc/libcuberille_prev.so (git-ignored, travels to the GPU box);
# then on the GPU box:  bash profiles/ab.sh run [ablate args]  alternates the two libraries three times.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "run" ]; then
  shift
  for i in 1 2 3; do
    echo -n "prev: "; CUBERILLE_LIB=$R/midas-journal-740_amd/csrc/libcuberille_prev.so python3 $R/profiles/ablate.py --reps 5 "$@" 2>/dev/null | cut -c1-150
    echo -n "tree: "; python3 $R/profiles/ablate.py --reps 5 "$@" 2>/dev/null | cut -c1-150
  done
  exit 0
fi
REV=${1:-HEAD}
T=$(mktemp -d)
mkdir -p $T/midas-journal-740_amd/csrc $T/include
for f in midas-journal-740_amd/csrc/cuberille_kernels.hip midas-journal-740_amd/csrc/cuberille_api.hip midas-journal-740_amd/csrc/cuberille_internal.h \
         midas-journal-740_amd/csrc/cuberille_vtk.cpp midas-journal-740_amd/csrc/Makefile include/cuberille_hip.h; do
  git -C $R show $REV:$f > $T/$f
done
make -s -C $T/midas-journal-740_amd/csrc
cp $T/midas-journal-740_amd/csrc/libcuberille_hip.so $R/midas-journal-740_amd/csrc/libcuberille_prev.so
rm -rf $T
echo "built $REV -> csrc/libcuberille_prev.so"
