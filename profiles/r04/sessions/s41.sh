cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04s41; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/avail.txt 2>&1
grep -o -E "\b(TA_[A-Z0-9_]+|TCP_[A-Z0-9_]+|TCC_[A-Z0-9_]*BUSY[A-Z0-9_]*|TD_[A-Z0-9_]+|TCC_REQ[A-Z0-9_]*|TCC_TAG_STALL[A-Z0-9_]*|TCC_NORMAL[A-Z0-9_]*|GRBM_[A-Z0-9_]+)\b" $O/avail.txt | sort -u | tr '\n' ' ' | cut -c1-6000
