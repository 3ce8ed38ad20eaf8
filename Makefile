# Top-level build: host index builder (g++), HIP engine (hipcc, gfx950), CPU oracle (gcc).
#   make -j8            everything (objects under build/, libraries in-tree next to the package so that they travel to the GPU box)
#   make test-libs      + the test-only engine build with 2^16-symbol rank superblocks (tests/test_gpu_large_index.py)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
CSRC = ema_amd/csrc
HOSTFLAGS = -O2 -g -fPIC -std=c++17 -Wall -Wextra -ffp-contract=off
HIPFLAGS = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(CSRC)
HOSTCLANG = -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(CSRC) -pthread

BWAABI = $(if $(wildcard $(CSRC)/bwaabi.cpp),ema_amd/libema_bwaabi.so)
all: ema_amd/libema_index.so ema_amd/libema_engine.so $(BWAABI) oracle
test-libs: all ema_amd/libema_engine_ss16.so

ema_amd/libema_index.so: $(CSRC)/index_build.cpp $(CSRC)/synth_genome.cpp
	$(CXX) $(HOSTFLAGS) -fopenmp -shared -o $@ $(CSRC)/index_build.cpp $(CSRC)/synth_genome.cpp -ldl -pthread

HIP_SRCS = $(wildcard $(CSRC)/*.hip)
HOST_SRCS = $(wildcard $(CSRC)/host_*.cpp)
ENGINE_HDRS = $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.hpp) $(wildcard include/*.h)
ENGINE_OBJS = $(patsubst $(CSRC)/%.hip,build/%.o,$(HIP_SRCS)) $(patsubst $(CSRC)/%.cpp,build/%.o,$(HOST_SRCS))
SS16_OBJS = $(patsubst $(CSRC)/%.hip,build/ss16/%.o,$(HIP_SRCS)) $(patsubst $(CSRC)/%.cpp,build/ss16/%.o,$(HOST_SRCS))

build/%.o: $(CSRC)/%.hip $(ENGINE_HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
build/%.o: $(CSRC)/%.cpp $(ENGINE_HDRS)
	@mkdir -p build
	$(HIPCC) $(HOSTCLANG) -c -o $@ $<
ema_amd/libema_engine.so: $(ENGINE_OBJS)
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o $@ $(ENGINE_OBJS)

# Test-only build of the same sources with rank superblocks of 2^16 symbols instead of 2^31, so that a 120 Kbp reference
# exercises the several-superblock branch of ema_lane_occ4 that only a human-size genome reaches otherwise.
build/ss16/%.o: $(CSRC)/%.hip $(ENGINE_HDRS)
	@mkdir -p build/ss16
	$(HIPCC) $(HIPFLAGS) -DEMA_OCC_SUPER_SHIFT=16 -c -o $@ $<
build/ss16/%.o: $(CSRC)/%.cpp $(ENGINE_HDRS)
	@mkdir -p build/ss16
	$(HIPCC) $(HOSTCLANG) -DEMA_OCC_SUPER_SHIFT=16 -c -o $@ $<
ema_amd/libema_engine_ss16.so: $(SS16_OBJS)
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o $@ $(SS16_OBJS)

# Development build with phase clocks in K3 / K4 (dev_prof.hpp; tools/gpu_k34_profile.py): `make prof-lib`
PROF_OBJS = $(patsubst $(CSRC)/%.hip,build/prof/%.o,$(HIP_SRCS)) $(patsubst $(CSRC)/%.cpp,build/prof/%.o,$(HOST_SRCS))
build/prof/%.o: $(CSRC)/%.hip $(ENGINE_HDRS)
	@mkdir -p build/prof
	$(HIPCC) $(HIPFLAGS) -DEMA_K34_PROF=1 -c -o $@ $<
build/prof/%.o: $(CSRC)/%.cpp $(ENGINE_HDRS)
	@mkdir -p build/prof
	$(HIPCC) $(HOSTCLANG) -DEMA_K34_PROF=1 -c -o $@ $<
ema_amd/libema_engine_prof.so: $(PROF_OBJS)
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o $@ $(PROF_OBJS)
prof-lib: ema_amd/libema_engine_prof.so

# A/B builds of the engine with extra compile-time flags: `make variant V=chainlds VFLAGS="-DEMA_CHAIN_REGS=0"` -> ema_amd/libema_engine_chainlds.so
# (EMA_ENGINE_LIB=libema_engine_chainlds.so selects it in the Python wrapper; tools/run_r05_ab.sh runs bench.py over a set of them)
VOBJS = $(patsubst $(CSRC)/%.hip,build/v_$(V)/%.o,$(HIP_SRCS)) $(patsubst $(CSRC)/%.cpp,build/v_$(V)/%.o,$(HOST_SRCS))
build/v_$(V)/%.o: $(CSRC)/%.hip $(ENGINE_HDRS)
	@mkdir -p build/v_$(V)
	$(HIPCC) $(HIPFLAGS) $(VFLAGS) -c -o $@ $<
build/v_$(V)/%.o: $(CSRC)/%.cpp $(ENGINE_HDRS)
	@mkdir -p build/v_$(V)
	$(HIPCC) $(HOSTCLANG) $(VFLAGS) -c -o $@ $<
variant: $(VOBJS)
	$(HIPCC) --offload-arch=gfx950 -fPIC -shared -o ema_amd/libema_engine_$(V).so $(VOBJS)

# libbwa-shaped face (include/ema_bwaabi.h): the 9 symbols the reference links from -lbwa, on top of the engine
ema_amd/libema_bwaabi.so: $(CSRC)/bwaabi.cpp include/ema_bwaabi.h include/ema_engine.h ema_amd/libema_engine.so
	$(CXX) $(HOSTFLAGS) -Iinclude -shared -o $@ $(CSRC)/bwaabi.cpp -Lema_amd -lema_engine -Wl,-rpath,'$$ORIGIN'

# oracle: the CPU checker; `ref` (only where /root/reference exists): the reference's own sources compiled where they lie into
# a directory OUTSIDE the repository ($EMA_REF_OUT, else $TMPDIR/ema_ref, else /tmp/ema_ref) -- util.c, count, preproc, the host half over the oracle's nine-symbol face (ema_refhost), and the same objects
# over the product's face (ema_ref_gpu, needs libema_bwaabi.so)
oracle: $(BWAABI)
	$(MAKE) -C oracle
	$(MAKE) -C oracle ref

clean:
	rm -rf build; rm -f ema_amd/*.so; $(MAKE) -C oracle clean

.PHONY: all test-libs prof-lib oracle clean variant
