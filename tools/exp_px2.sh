#!/bin/bash
echo base; python tools/px_time.py
for v in NO_BALLOT NO_CONV NO_POOL NO_BOTH; do echo $v; PH_ALT_LIB=tools/libpolyhead_px_$v.so python tools/px_time.py | grep poolx; done
