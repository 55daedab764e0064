# Convenience targets (the driver uses __graft_entry__.build / pytest / bench.py directly).
.PHONY: build test-cpu test-gpu bench clean asan
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

# CPU sanitizer run (build container only, never on the GPU box): AddressSanitizer +
# UndefinedBehaviorSanitizer builds of the C++ host mirror (TOML reader, scenario, field builder /
# FMM, simulator, flattened C API) and of the oracle, then the CPU suite on top of them.  The HIP
# library itself stays the normal build (its host code is hipcc's; GPU ASan is not available here).
ASAN_DIR := build/asan
ASAN_FLAGS := -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined
asan: build
	mkdir -p $(ASAN_DIR)
	g++ $(ASAN_FLAGS) -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wall -Wextra -pthread \
	    -Iinclude -Ipedoni_amd/csrc/host -o $(ASAN_DIR)/libpedoni_host.so \
	    $(filter-out %_main.cpp,$(wildcard pedoni_amd/csrc/host/*.cpp)) -Lpedoni_amd/lib -lpedoni_hip \
	    -Wl,-rpath,$(abspath pedoni_amd/lib)
	gcc $(ASAN_FLAGS) -std=gnu11 -fPIC -shared -ffp-contract=off -fno-fast-math -mfma -fopenmp -Wall -Wextra \
	    -o $(ASAN_DIR)/libpedoni_oracle.so $(addprefix oracle/,oracle_util.c oracle_field.c oracle_sfm.c oracle_sim.c) -lm
	LD_PRELOAD="$$(gcc -print-file-name=libasan.so) $$(gcc -print-file-name=libubsan.so)" \
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	PEDONI_HOST_LIB=$(abspath $(ASAN_DIR)/libpedoni_host.so) PEDONI_ORACLE_LIB=$(abspath $(ASAN_DIR)/libpedoni_oracle.so) \
	python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tee $(ASAN_DIR)/pytest.log
