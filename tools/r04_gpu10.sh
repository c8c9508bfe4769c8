#!/bin/bash
NEEDLE_HIP_TRACE=1 python -c "
from needle_amd import capi
capi.set_device(0)
print(capi.int_valu_ceiling())" 2>&1 | grep -a "ceiling\|e+" 
