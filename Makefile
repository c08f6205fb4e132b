# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
PYTHON ?= python

all: build

build:
	$(MAKE) -C mp-mvs_amd/csrc
	$(MAKE) -C mp-mvs_amd/host
	$(MAKE) -C oracle

test-cpu: build
	$(PYTHON) -m pytest tests -q -m "not gpu"

test-gpu: build
	$(PYTHON) -m pytest tests -q -m gpu

bench: build
	$(PYTHON) bench.py

clean:
	$(MAKE) -C mp-mvs_amd/csrc clean
	$(MAKE) -C mp-mvs_amd/host clean
	$(MAKE) -C oracle clean

.PHONY: all build test-cpu test-gpu bench clean
