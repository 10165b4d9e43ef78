export XCLIP_PROBE_NO_REF=1
for b in 0 1500 3000 4500; do echo "bwd stagger $b"; XCLIP_ATTN_STAGGER_BWD=$b python tools/probe_attn_abl.py 2>&1 | grep ATTN; done
for f in 0 600 1300 2000; do echo "fwd stagger $f"; XCLIP_ATTN_STAGGER_FWD=$f python tools/probe_perf.py 2>&1 | grep "attention fwd"; done
