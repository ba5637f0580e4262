# Builds the reference's OWN LZ4 library from its sources where they lie (contrib/libs/lz4, contrib/libs/xxhash: plain C,
# no generated code, no build system needed) into oracle/_ref/liblz4_ref.so.  Used only to validate oracle/lz4_oracle.c
# and the GPU decoder in this container; /root/reference does not exist on the GPU box.
REF ?= /root/reference
CC ?= gcc
OUT = _ref/liblz4_ref.so
SRCS = $(REF)/contrib/libs/lz4/lz4.c $(REF)/contrib/libs/lz4/lz4frame.c $(REF)/contrib/libs/lz4/lz4hc.c $(REF)/contrib/libs/xxhash/xxhash.c

$(OUT): $(SRCS)
	@mkdir -p _ref
	$(CC) -O2 -fPIC -shared -I$(REF)/contrib/libs/lz4 -I$(REF)/contrib/libs/xxhash -o $@ $(SRCS)
