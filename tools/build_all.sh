#!/bin/bash
# build the product library and the stamped diagnostic variant, from wherever this is called
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py | tail -1
GHF_VARIANT=stamps python -c "
import sys; sys.path.insert(0,'.')
from graph_hypernetwork_forge_amd import _build; print(_build.build())" | tail -1
GHF_VARIANT=ablate python -c "
import sys; sys.path.insert(0,'.')
from graph_hypernetwork_forge_amd import _build; print(_build.build())" | tail -1
