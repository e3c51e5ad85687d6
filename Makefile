# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
build:
	python -c "import __graft_entry__ as g; g.build()"
test:
	python -m pytest tests -x -q -m "not gpu"
test-gpu:
	python -m pytest tests -x -q -m gpu
bench:
	python bench.py
clean:
	$(MAKE) -C quade_amd/csrc clean
	$(MAKE) -C oracle clean
.PHONY: build test test-gpu bench clean
