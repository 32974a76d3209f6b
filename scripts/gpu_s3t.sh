#!/bin/bash
F='grep -v "Warn\|warn\|amdgpu\|Consider\|self.losses\|^-\|Network\|^$"'
for o in "" "batch_query_passes=0" "reuse_key_features=0" "nce_sequential_keys=1" "overlap_registration=0" "batch_query_passes=0,reuse_key_features=0,nce_sequential_keys=1,overlap_registration=0"; do
  OPTS=$o python scripts/diag_first_update.py 2>&1 | grep -E "^opts|^G.model|^F.mlp|^R.flow|^total|^parameter"
done
DFMIR_NO_NCE_FUSED=1 python scripts/diag_first_update.py 2>&1 | grep -E "^opts|^G.model|^F.mlp|^R.flow|^total"
