#!/bin/bash
# Writes .git_rev = "<short HEAD>[-dirty]" for the GPU box (which gets the tree without .git): tools/profile_round.sh records it as
# `rev` next to the content hash `build` of the library that was profiled.  "-dirty": tracked files differ from HEAD, i.e. `rev`
# names the commit BEFORE the one that will hold these sources; `build` is what ties a profile to a library either way.
cd "$(dirname "$0")/.."
rev=$(git rev-parse --short HEAD)
if [ -n "$(git status --porcelain --untracked-files=no)" ]; then rev="$rev-dirty"; fi
echo "$rev" > .git_rev
cat .git_rev
