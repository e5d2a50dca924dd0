# Top-level build: the HIP library (gfx950), the checker, and the C++ host application.
#   make            everything
#   make lib        differentiable-renderer_amd/libdrt_hip.so   (hipcc --offload-arch=gfx950)
#   make host       build/render   (src/render_hip.cpp against include/drt + libdrt_hip.so)
#   make oracle     oracle/libdrt_oracle.so (+ oracle/_ref where /root/reference exists)
PKG := differentiable-renderer_amd
LIB := $(PKG)/libdrt_hip.so
CXX ?= g++
HIPCC ?= hipcc
# the EXR writer compresses (ZIP blocks) when zlib is there, writes uncompressed scan lines otherwise
# (no '#' inside $(shell ...): GNU Make >= 4.3 hands "\#" to the shell verbatim and the probe always "succeeds")
# DRT_NO_ZLIB=1 forces the uncompressed fallback (tests/test_host_api.py builds both ways)
EXR_ZLIB := $(if $(DRT_NO_ZLIB),,$(shell echo 'int main(){return 0;}' | $(CXX) -x c++ -include zlib.h -fsyntax-only - >/dev/null 2>&1 && echo -DDRT_EXR_ZLIB))
EXR_ZLIB_LIB := $(if $(EXR_ZLIB),-lz)

all: lib oracle host

lib: $(LIB)
$(LIB): $(PKG)/csrc/drt_hip.hip $(wildcard $(PKG)/csrc/*.h) include/drt_hip.h
	python3 $(PKG)/csrc/embed_sources.py
	$(HIPCC) --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -shared -Iinclude -o $@ $(PKG)/csrc/drt_hip.hip -lrccl -lhiprtc

oracle:
	$(MAKE) -C oracle libdrt_oracle.so ref

host: build/render
build/render: src/render_hip.cpp src/args.hpp src/write.hpp $(wildcard include/drt/*.hpp) include/drt_hip.h $(LIB)
	mkdir -p build
	$(CXX) -O3 -std=c++17 -Iinclude -Isrc $(EXR_ZLIB) -o $@ src/render_hip.cpp -L$(PKG) -ldrt_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)' -lpthread $(EXR_ZLIB_LIB)

clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all lib oracle host clean
