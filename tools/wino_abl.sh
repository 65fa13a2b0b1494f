for a in 0 1 2 4 8 6 14 13 11 7 15; do echo "ABL $a"; MVSGI_WINO_ABL=$a timeout -k 10 120 python3 tools/wino_probe.py --shape 64 8 40 160 2>&1 | grep "^direct" | tail -1; done
