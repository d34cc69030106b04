for pad in 0 32 544 4128; do echo "pad $pad"; RF_RSEG_PAD_CELLS=$pad timeout -k 10 120 python3 tools/chunk_fwd.py 0 2>&1 | tail -1; done
