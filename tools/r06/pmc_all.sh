# round 6: counter passes behind profiles/stft_pmc.json (all five launches bench.py's roofline entries time)
bash tools/pmc_stft.sh r06/pmc1024 1024 1024 44100 > /dev/null 2>&1
bash tools/pmc_stft.sh r06/pmc4096 4096 32 1323000 > /dev/null 2>&1
bash tools/pmc_stft.sh r06/pmcnfk1024 1024 1024 44100 RUNNER=tools/r04/run_nfk_only.py > /dev/null 2>&1
bash tools/pmc_stft.sh r06/pmcnfk4096 4096 32 1323000 RUNNER=tools/r04/run_nfk_only.py > /dev/null 2>&1
bash tools/pmc_stft.sh r06/pmcmel 0 1024 0 RUNNER=tools/r05/run_mel_only.py MATCH=mel_ > /dev/null 2>&1
for d in pmc1024 pmc4096 pmcnfk1024 pmcnfk4096 pmcmel; do echo "== $d"; cat gpurun_out/r06/$d/summary.txt | head -8; done
