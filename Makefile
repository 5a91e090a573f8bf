# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
all:
	python3 -c "import __graft_entry__ as g; g.build()"
test:
	python3 -m pytest tests -x -q -m "not gpu"
test-gpu:
	python3 -m pytest tests -x -q -m gpu
bench:
	python3 bench.py
clean:
	$(MAKE) -C qo-100-tools_amd/csrc clean
	$(MAKE) -C qo-100-tools_amd/host clean
	$(MAKE) -C oracle clean
	$(MAKE) -C tools clean
.PHONY: all test test-gpu bench clean
