# Convenience targets (the driver uses __graft_entry__.build / pytest / bench.py directly).
.PHONY: build test-cpu test-gpu bench clean
build:
	python -m pedoni_amd.build
test-cpu: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py --gpus 1 --steps 100 --warmup 10
clean:
	rm -rf pedoni_amd/lib pedoni_amd/bin oracle/libpedoni_oracle.so
