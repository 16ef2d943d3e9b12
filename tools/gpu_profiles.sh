#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
bash tools/profile_round.sh r03i || exit 1
bash tools/profile_round.sh r03i5 --config 5 || exit 1
bash tools/profile_round.sh r03i4 --config 4 || exit 1
