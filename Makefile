# Build libhosrender.so (gfx950 only) and nothing else.  `python -c "import __graft_entry__ as g; g.build()"` calls this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := $(wildcard hosnerf_amd/csrc/*.hip)
OBJ   := $(patsubst hosnerf_amd/csrc/%.hip,build/%.o,$(SRC))
LIB   := hosnerf_amd/lib/libhosrender.so
# -packed-fp32-ops (device target feature OFF): no v_pk_{mul,add,fma}_f32 / v_pk_mov_b32 is ever emitted.  Measured on gfx950 /
# ROCm 7.2 (scripts/stress_victims.py): a wave executing packed-FP32 VALU instructions computes WRONG values in
# lanes 48-63 when waves of ANOTHER kernel that issue MFMAs are co-resident on the same SIMD (two HIP streams).  Without the
# packed forms the same kernels are bit-stable under any concurrency; the step time is unchanged (the VALU work here is not
# what bounds any kernel).  tests/test_isa_hazards_cpu.py checks the built ISA for stragglers.
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude -Ihosnerf_amd/csrc -Wno-unused-result

COMM  := hosnerf_amd/lib/libhoscomm.so
ROCM_PATH ?= $(shell d=$$(dirname $$(dirname $$(readlink -f $(HIPCC)))); [ -d $$d/include ] && echo $$d || echo /opt/rocm)
HAVE_RCCL := $(shell [ -f $(ROCM_PATH)/include/rccl/rccl.h ] && ls $(ROCM_PATH)/lib/librccl.so* >/dev/null 2>&1 && echo 1)

# libhoscomm.so (in-graph collectives, optional: torch.distributed stays the default path) is built only where RCCL exists
all: $(LIB) $(if $(HAVE_RCCL),$(COMM),comm-skipped)

comm-skipped:
	@echo "note: rccl.h / librccl.so not found under $(ROCM_PATH): libhoscomm.so not built (hosnerf_amd.comm falls back to torch.distributed)"

build/%.o: hosnerf_amd/csrc/%.hip hosnerf_amd/csrc/hos_common.h hosnerf_amd/csrc/hos_gemm_common.h include/hosrender.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p hosnerf_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

# the collectives of the data-parallel step as a C ABI over RCCL (include/hoscomm.h); host code only, its own library so that
# libhosrender.so does not depend on librccl.so
$(COMM): hosnerf_amd/csrc_comm/hos_comm.cpp include/hoscomm.h
	@mkdir -p hosnerf_amd/lib
	$(HIPCC) -O2 -std=c++17 -fPIC -shared -Iinclude -I$(ROCM_PATH)/include $< -o $@ -L$(ROCM_PATH)/lib -lrccl -Wl,-rpath,$(ROCM_PATH)/lib

clean:
	rm -rf build $(LIB) $(COMM)
.PHONY: all clean comm-skipped
