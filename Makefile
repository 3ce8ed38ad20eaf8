# Top-level build: host index builder (g++), HIP engine (hipcc, gfx950), CPU oracle (gcc).
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
CSRC = ema_amd/csrc
HOSTFLAGS = -O2 -g -fPIC -std=c++17 -Wall -Wextra -ffp-contract=off

all: ema_amd/libema_index.so ema_amd/libema_engine.so oracle

ema_amd/libema_index.so: $(CSRC)/index_build.cpp
	$(CXX) $(HOSTFLAGS) -fopenmp -shared -o $@ $<

ENGINE_SRCS = $(wildcard $(CSRC)/*.hip) $(CSRC)/host_index.cpp $(CSRC)/host_append.cpp $(CSRC)/host_ingest.cpp $(CSRC)/host_sam.cpp
ENGINE_HDRS = $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hpp) include/ema_engine.h include/ema_ingest.h include/ema_sam.h
ema_amd/libema_engine.so: $(ENGINE_SRCS) $(ENGINE_HDRS)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude -I$(CSRC) -o $@ $(ENGINE_SRCS)

oracle:
	$(MAKE) -C oracle
	$(MAKE) -C oracle ref

clean:
	rm -f ema_amd/*.so; $(MAKE) -C oracle clean

.PHONY: all oracle clean
