# Builds
#   vqgan-training_amd/libvqhip.so   the product: HIP kernels for gfx950 behind the C ABI (include/vqhip.h)
#   tests/emu/libvqhip_emu.so        TEST ONLY: the same kernel sources compiled for host cores through
#                                    tests/emu/hip_emu.h (fiber emulation; no GPU needed)
#   oracle/_ref/libvq_oracle.so      TEST ONLY: the C restatement of the VQ lookup (bit-exact oracle)
HIPCC ?= /opt/rocm/bin/hipcc
HOSTCXX ?= /opt/rocm/lib/llvm/bin/clang++
CC ?= gcc
ARCH ?= gfx950
# make ABLATE=1: also compile the profiling-only ablated kernel copies (no DMA / no MFMA)
ABLATE_FLAGS := $(if $(ABLATE),-DVQ_ABLATION_KERNELS,)
# -fno-slp-vectorize: hipcc's SLP pass packs the epilogues' scalar fp32 adds / multiplies into v_pk_add_f32 / v_pk_mul_f32, which cost
# MORE than the scalar pair beside MFMAs on gfx950 (MI355X_MICROARCH.md); measured on the 128-channel bf16 layers +2-3 %, step +0.1 %
# (profiles/r3k_variants_micro.txt, r3k_bench_ab.txt)
# -fvisibility=hidden: only the entry points include/vqhip.h declares (inside its `visibility push(default)`) are dynamic symbols
HIPOPT := -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value -fno-slp-vectorize

CSRC := vqgan-training_amd/csrc
KERNELS := $(CSRC)/conv_igemm.hip $(CSRC)/conv_wgrad.hip $(CSRC)/gn_silu.hip $(CSRC)/layout_pool.hip \
           $(CSRC)/loss_ops.hip $(CSRC)/optim_vq.hip $(CSRC)/debug_probe.hip $(CSRC)/conv_small.hip $(CSRC)/image_ops.hip $(CSRC)/attention.hip
HDRS := $(CSRC)/vq_common.h include/vqhip.h

LIB := vqgan-training_amd/libvqhip.so
EMU := tests/emu/libvqhip_emu.so
ORACLE := oracle/_ref/libvq_oracle.so

OBJS := $(patsubst $(CSRC)/%.hip,build/hip/%.o,$(KERNELS))
EMUOBJS := $(patsubst $(CSRC)/%.hip,build/emu/%.o,$(KERNELS))

all: $(LIB)
emu: $(EMU)
oracle: $(ORACLE)

build/hip/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/hip
	$(HIPCC) --offload-arch=$(ARCH) $(HIPOPT) $(ABLATE_FLAGS) -c $< -o $@

build/hip/capi_common.o: $(CSRC)/capi_common.cpp include/vqhip.h
	@mkdir -p build/hip
	$(HOSTCXX) -O2 -std=c++17 -fPIC -fvisibility=hidden -c $< -o $@

$(LIB): $(OBJS) build/hip/capi_common.o $(CSRC)/libvqhip.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--version-script=$(CSRC)/libvqhip.map -o $@ $(filter %.o,$^)

# (the emulator library is test infrastructure: built with the ablation knobs on, so that the CPU suite reaches them; cycle stamps excepted)
build/emu/%.o: $(CSRC)/%.hip $(HDRS) tests/emu/hip_emu.h
	@mkdir -p build/emu
	$(HOSTCXX) -x c++ -O2 -std=c++17 -fPIC -Itests/emu -include tests/emu/hip_emu.h -Wno-unused-value -DVQ_ABLATION_KERNELS -c $< -o $@

build/emu/emu_rt.o: tests/emu/emu_rt.cpp tests/emu/hip_emu.h
	@mkdir -p build/emu
	$(HOSTCXX) -O2 -std=c++17 -fPIC -Itests/emu -c $< -o $@

build/emu/capi_common.o: $(CSRC)/capi_common.cpp include/vqhip.h
	@mkdir -p build/emu
	$(HOSTCXX) -O2 -std=c++17 -fPIC -c $< -o $@

$(EMU): $(EMUOBJS) build/emu/emu_rt.o build/emu/capi_common.o
	$(HOSTCXX) -shared -fPIC -o $@ $^ -lpthread

$(ORACLE): oracle/vq_oracle.c
	@mkdir -p oracle/_ref
	$(CC) -O2 -std=c99 -ffp-contract=off -shared -fPIC -o $@ $< -lm

# make ablate: TOOLS ONLY — the library with the profiling ablations (no-DMA / no-MFMA copies, epilogue pricing, VqGnBwdFuse) compiled in
# (VQ_ABLATION_KERNELS), as a SEPARATE file that only tools/ load explicitly ($VQ_ABLATE_LIB)
ABLATE_LIB := build/ablate/libvqhip_ablate.so
ABLOBJS := $(patsubst $(CSRC)/%.hip,build/ablate/%.o,$(KERNELS))
build/ablate/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/ablate
	$(HIPCC) --offload-arch=$(ARCH) $(HIPOPT) -DVQ_ABLATION_KERNELS -c $< -o $@
$(ABLATE_LIB): $(ABLOBJS) build/hip/capi_common.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^
ablate: $(ABLATE_LIB)

clean:
	rm -rf build $(LIB) $(EMU) oracle/_ref

.PHONY: all emu oracle ablate clean
