#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "--no-other-configs --no-profile --steps 20" "--no-other-configs --no-profile --steps 20 --interp linear" "--no-other-configs --no-profile --steps 80 --dots 25" "--no-other-configs --no-profile --steps 200 --dots 25 --interp linear"; do
  tools/ab.sh "$cfg" r04 default r04 default 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12,$13,$14,$15,$16,$17,$18,$19,$20}'
done
