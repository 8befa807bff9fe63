#!/bin/bash
# the SQ counter passes of pmc_quick.sh for two builds of the library:  tools/scripts/pmc_ab.sh <libA> <libB> <run_variant.py args...>
: ${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it): an empty root would turn cd / rm -rf below into operations on /}
set -e
R=$GRAFT_REPO_ROOT
A=$1; B=$2; shift; shift
export FLAN_AMD_LIB=$R/$A
bash $R/tools/scripts/pmc_quick.sh libA "$@"
export FLAN_AMD_LIB=$R/$B
bash $R/tools/scripts/pmc_quick.sh libB "$@"
