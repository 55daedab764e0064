#!/bin/bash
# Before `gpurun -- bash tools/ab_prev.sh TAG`: put the sources of an earlier commit where the GPU box (which has no .git)
# can build them:  bash tools/ab_prev_prepare.sh [REV]     (exp/ is git-ignored but travels with the snapshot)
REV=${1:-HEAD}
mkdir -p exp/prev_csrc
for f in $(git ls-tree --name-only $REV pedoni_amd/csrc/ | grep -E '\.(hpp|hip)$'); do git show $REV:$f > exp/prev_csrc/$(basename $f); done
ls exp/prev_csrc
