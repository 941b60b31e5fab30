# Repository-level entry points (the library itself: make -C hicpeaks_amd/csrc).
ROUND ?= r06
GPURUN ?= /usr/local/graft/bin/gpurun

lib:
	$(MAKE) -C hicpeaks_amd/csrc

# the round's measurements on a GPU box -> gpurun_out/$(ROUND)/, summaries -> profiles/$(ROUND)_* (scripts/README.md, profiles/README.md)
profiles: lib
	$(GPURUN) --timeout 3000 -- 'bash scripts/measure/profile_round.sh $(ROUND)'
	python scripts/measure/collect_profiles.py $(ROUND)

test:
	python -m pytest tests -x -q -m "not gpu"

.PHONY: lib profiles test
