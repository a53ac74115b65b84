# Builds libofx_hip.so (HIP kernels + C ABI, gfx950).  The oracle is Python (oracle/*.py): nothing of it is compiled.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := outfitx_amd/csrc
OBJ   := build/obj
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(SRCS))
LIB   := outfitx_amd/libofx_hip.so
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $(if $(DIAG),-DOFX_DIAG,) $(EXTRA)

all: $(LIB)

$(OBJ)/%.o: $(CSRC)/%.hip $(CSRC)/ofx_common.h $(CSRC)/gemm_common.h include/ofx.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB)
.PHONY: all clean
