#!/bin/bash
# Round 5, closing lease: the driver's sequence verbatim + trace / PMC passes of the final library
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
bash tools/r3_driver_verbatim.sh r5_final
bash tools/r5_profile.sh
