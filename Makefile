# Builds the gfx950 product library (hipcc cross-compiles without a GPU) and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
PKG   := multiple-quadrotor-slam_amd
SRCS  := $(wildcard $(PKG)/csrc/*.hip)
HDRS  := $(wildcard $(PKG)/csrc/*.h) include/mqslam.h
LIB   := $(PKG)/libmqslam_hip.so
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wl,-rpath,/opt/rocm/lib
LIBS     ?= -L/opt/rocm/lib -pthread

all: $(LIB) oracle

# one object per translation unit (kernels never call across files), so a change recompiles one file and `make -j` uses the cores
OBJDIR := build/obj
OBJS   := $(patsubst $(PKG)/csrc/%.hip,$(OBJDIR)/%.o,$(SRCS))

$(OBJDIR)/%.o: $(PKG)/csrc/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(OBJS) $(LIBS)

oracle:
	$(MAKE) -s -C oracle/c

tests/libhost_math.so: tests/host_math.cpp $(PKG)/csrc/tri_math.h $(PKG)/csrc/ba_math.h $(PKG)/csrc/cam_math.h $(PKG)/csrc/pnp_math.h
	g++ -O2 -ffp-contract=off -fPIC -shared -o $@ tests/host_math.cpp

clean:
	rm -f $(LIB) tests/libhost_math.so
	$(MAKE) -s -C oracle/c clean

.PHONY: all oracle clean
