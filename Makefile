# Build libhosrender.so (gfx950 only) and nothing else.  `python -c "import __graft_entry__ as g; g.build()"` calls this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := $(wildcard hosnerf_amd/csrc/*.hip)
OBJ   := $(patsubst hosnerf_amd/csrc/%.hip,build/%.o,$(SRC))
LIB   := hosnerf_amd/lib/libhosrender.so
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Iinclude -Ihosnerf_amd/csrc -Wno-unused-result

all: $(LIB)

build/%.o: hosnerf_amd/csrc/%.hip hosnerf_amd/csrc/hos_common.h hosnerf_amd/csrc/hos_gemm_common.h include/hosrender.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p hosnerf_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

clean:
	rm -rf build $(LIB)
.PHONY: all clean
