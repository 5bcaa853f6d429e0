#!/bin/bash
# Evidence, part 1 (GPU box): headline profile set and the other shapes.   bash tools/r03_evidence1.sh r03_a
tag=${1:-r03_a}
bash tools/profile_round.sh $tag
bash tools/profile_shapes.sh $tag
