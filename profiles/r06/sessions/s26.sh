#!/bin/bash
# r06 session 26: the round's evidence on the final tree (after the equal cut and the 22-bit block numbers of the one-word builds):
#   gpurun -- 'bash profiles/r06/collect_r06.sh r06x'
bash profiles/r06/collect_r06.sh r06x
