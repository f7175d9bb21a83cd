/*
 * GpuBVGraph -- the reference-side binding of libbvgpu: an ImmutableGraph whose successor lists are decoded on an
 * MI355X.  Drop-in through the reference's own plug-in mechanism: put
 *     graphclass=it.unimi.dsi.webgraph.gpu.GpuBVGraph
 * in <basename>.properties (or pass `-g GpuBVGraph` to the tools that take a graph class), and
 * ImmutableGraph.load(basename) reflects on the static load methods below (ImmutableGraph.java:647-685; the
 * required signatures are listed at ImmutableGraph.java:89-104).  Same .graph/.offsets/.properties files.
 *
 * NOT COMPILED IN THIS REPOSITORY: the build image has no JDK (no javac, no jni.h) and the reference's
 * dependencies (dsiutils, fastutil) are not vendored.  Shipped as the source a maintainer would add; the C side
 * is java/jni/bvgpu_jni.c.  The same surface is exercised here through webgraph_amd/host/bvgraph.hpp (C++) and
 * webgraph_amd/bvgraph.py (ctypes).
 */
package it.unimi.dsi.webgraph.gpu;

import it.unimi.dsi.logging.ProgressLogger;
import it.unimi.dsi.webgraph.ImmutableGraph;
import it.unimi.dsi.webgraph.LazyIntIterator;
import it.unimi.dsi.webgraph.LazyIntIterators;
import it.unimi.dsi.webgraph.NodeIterator;

import java.io.IOException;
import java.util.NoSuchElementException;

public class GpuBVGraph extends ImmutableGraph {
	static { System.loadLibrary("bvgpu_jni"); }

	/** Nodes decoded per GPU call by a sequential iterator. */
	public static final int BATCH_NODES = 1 << 20;

	private final long handle; // bvg_t*
	private final CharSequence basename;
	private final int n, windowSize, maxRefCount;
	private final long m;

	// ---- natives: one per entry point of include/bvgpu.h; a negative bvg_status becomes the exception the
	// reference throws at the same place (BVG_EARG -> IllegalArgumentException, BVG_ESTATE -> IllegalStateException,
	// BVG_EUNSUPPORTED -> UnsupportedOperationException, BVG_EIO -> IOException wrapped in RuntimeException).
	private static native long open(String basename, int device) throws IOException;           // bvg_open
	private static native long cloneHandle(long handle);                                        // bvg_clone
	private static native void close(long handle);                                              // bvg_close
	private static native long[] info(long handle);                                             // bvg_info: {nodes, arcs, window, maxref}
	private static native int outdegree(long handle, int x);                                    // bvg_outdegrees(x, x+1)
	private static native int[] successorArray(long handle, int x);                             // bvg_successors_batch, q = 1
	/** Fills rowptr[to-from+1]; returns the successors of nodes [from,to) concatenated.  ONE native call: the results arrive in
	 *  the handle's pinned buffers (chunks cross PCIe while the next chunk is decoded) and are copied into a fresh int[]. */
	private static native int[] decodeRange(long handle, int from, int to, long[] rowptr);      // bvg_decode_range_view
	/** hashCode() continued from h over [from,to) on the device; nothing is materialised. */
	private static native int scanChecksum(long handle, int from, int to, int h);               // bvg_scan_checksum
	private static native boolean equalRange(long handleA, long handleB, int from, int to);      // bvg_equal_range
	/** bvg_store: compresses the CSR (rowptr[n+1], succ) on the device and writes basename.graph / .offsets / .properties. */
	private static native void storeCsr(String basename, int device, int n, long[] rowptr, int[] succ, int windowSize, int maxRefCount, int minIntervalLength,
		int zetaK, int flags, int numberOfThreads) throws IOException;
	/** bvg_recompress: decode -> compress without leaving the device. */
	private static native void recompress(long handle, String basename, int windowSize, int maxRefCount, int minIntervalLength, int zetaK, int flags,
		int numberOfThreads) throws IOException;

	private GpuBVGraph(final long handle, final CharSequence basename) {
		this.handle = handle;
		this.basename = basename;
		final long[] i = info(handle);
		n = (int)i[0]; m = i[1]; windowSize = (int)i[2]; maxRefCount = (int)i[3];
	}

	// ---- the loaders ImmutableGraph.load reflects on
	public static GpuBVGraph load(final CharSequence basename) throws IOException { return new GpuBVGraph(open(basename.toString(), 0), basename); }
	public static GpuBVGraph load(final CharSequence basename, final ProgressLogger pl) throws IOException { return load(basename); }
	public static GpuBVGraph loadMapped(final CharSequence basename) throws IOException { return load(basename); }
	public static GpuBVGraph loadMapped(final CharSequence basename, final ProgressLogger pl) throws IOException { return load(basename); }
	public static GpuBVGraph loadOffline(final CharSequence basename) throws IOException { return load(basename); }
	public static GpuBVGraph loadOffline(final CharSequence basename, final ProgressLogger pl) throws IOException { return load(basename); }

	// ---- BVGraph.store (BVGraph.java:1679-1730): same arguments, same files, compressed on the GPU
	public static void store(final ImmutableGraph graph, final CharSequence basename, final int windowSize, final int maxRefCount, final int minIntervalLength,
			final int zetaK, final int flags, final int numberOfThreads, final ProgressLogger pl) throws IOException {
		if (graph instanceof GpuBVGraph) { recompress(((GpuBVGraph)graph).handle, basename.toString(), windowSize, maxRefCount, minIntervalLength, zetaK, flags, numberOfThreads); return; }
		// any other ImmutableGraph: drain it into a CSR (what storeInternal's node iterator does, BVGraph.java:2471-2550) and hand that over
		final int n = graph.numNodes();
		final long[] rowptr = new long[n + 1];
		int[] succ = new int[(int)Math.min(Integer.MAX_VALUE - 8, Math.max(16, graph.numArcs() >= 0 ? graph.numArcs() : 16))];
		long m = 0;
		final NodeIterator it = graph.nodeIterator();
		for (int x = 0; x < n; x++) {
			it.nextInt();
			final int d = it.outdegree();
			final int[] s = it.successorArray();
			if (m + d > Integer.MAX_VALUE - 8) throw new UnsupportedOperationException("more than 2^31 arcs: store the graph in parts");
			if (m + d > succ.length) succ = java.util.Arrays.copyOf(succ, (int)Math.min(Integer.MAX_VALUE - 8, Math.max(m + d, succ.length * 2L)));
			System.arraycopy(s, 0, succ, (int)m, d);
			m += d;
			rowptr[x + 1] = m;
		}
		storeCsr(basename.toString(), 0, n, rowptr, succ, windowSize, maxRefCount, minIntervalLength, zetaK, flags, numberOfThreads);
	}
	public static void store(final ImmutableGraph graph, final CharSequence basename, final int windowSize, final int maxRefCount, final int minIntervalLength,
			final int zetaK, final int flags) throws IOException { store(graph, basename, windowSize, maxRefCount, minIntervalLength, zetaK, flags, 1, null); }
	/** BVGraph.store(graph, basename) with the reference's defaults (BVGraph.java:455-470: window 7, maxRefCount 3, minIntervalLength 4, zeta_3). */
	public static void store(final ImmutableGraph graph, final CharSequence basename) throws IOException { store(graph, basename, 7, 3, 4, 3, 0); }

	@Override public int numNodes() { return n; }
	@Override public long numArcs() { return m; }
	@Override public boolean randomAccess() { return true; }
	@Override public boolean hasCopiableIterators() { return true; }
	@Override public CharSequence basename() { return basename; }
	public int windowSize() { return windowSize; }
	public int maxRefCount() { return maxRefCount; }

	@Override public int outdegree(final int x) {
		if (x < 0 || x >= n) throw new IllegalArgumentException("Node index out of range: " + x);
		return outdegree(handle, x);
	}
	@Override public int[] successorArray(final int x) {
		if (x < 0 || x >= n) throw new IllegalArgumentException("Node index out of range: " + x);
		return successorArray(handle, x);
	}
	@Override public LazyIntIterator successors(final int x) {
		final int[] a = successorArray(x);
		return LazyIntIterators.wrap(a, a.length);
	}
	/** Flyweight copy sharing the staged graph (BVGraph.copy()). */
	@Override public GpuBVGraph copy() { return new GpuBVGraph(cloneHandle(handle), basename); }

	@Override public NodeIterator nodeIterator(final int from) { return new BatchIterator(handle, false, from, Integer.MAX_VALUE); }

	/** ImmutableGraph.hashCode() (ImmutableGraph.java:757-770) as one checksum scan on the device. */
	@Override public int hashCode() { return scanChecksum(handle, 0, n, -1); }
	/** ImmutableGraph.equals (ImmutableGraph.java:731-749); two graphs of this class are compared on the device, row by row, without a list reaching the JVM. */
	@Override public boolean equals(final Object o) {
		if (o instanceof GpuBVGraph) {
			final GpuBVGraph g = (GpuBVGraph)o;
			if (n != g.n) return false;
			// (handles on different devices, or shard handles that stage different ranges: bvg_equal_range answers BVG_EARG, which the binding throws as
			// IllegalArgumentException -- then the comparison of ImmutableGraph.equals through the iterators, as the C++ and Python mirrors do)
			try { return equalRange(handle, g.handle, 0, n); } catch (final IllegalArgumentException e) { return super.equals(o); }
		}
		return super.equals(o);
	}

	/** Sequential scan served from GPU-decoded batches; same contract as BVGraph's node iterator. */
	private final class BatchIterator extends NodeIterator {
		private final int from, limit;
		private int curr, lo, hi;
		private long[] rowptr;
		private int[] succ;
		/** The bvg_t this iterator decodes through.  A copy (and so every split iterator) owns a flyweight handle of its
		 *  own (bvg_clone): the reference hands copies to other threads (BVGraph.java:2471-2477, ImmutableGraph.java:379-409)
		 *  and a bvg_t is not thread-safe.  Closed when the iterator is exhausted, or by the finalizer. */
		private long h;
		private final boolean ownsHandle;

		BatchIterator(final long h, final boolean ownsHandle, final int from, final int upperBound) {
			if (from < 0 || from > n) throw new IllegalArgumentException("Node index out of range: " + from);
			this.h = h; this.ownsHandle = ownsHandle;
			this.from = from; curr = from - 1; lo = hi = from;
			limit = Math.min(upperBound, n) - 1;
		}
		private void release() { if (ownsHandle && h != 0) { close(h); h = 0; } }
		@SuppressWarnings("deprecation")
		@Override protected void finalize() throws Throwable { try { release(); } finally { super.finalize(); } }
		@Override public boolean hasNext() { return curr < limit; }
		@Override public int nextInt() {
			if (!hasNext()) throw new NoSuchElementException();
			if (++curr >= hi) {
				lo = curr; hi = (int)Math.min((long)lo + BATCH_NODES, (long)limit + 1);
				rowptr = new long[hi - lo + 1];
				succ = decodeRange(h, lo, hi, rowptr);
				if (hi > limit) release(); // the last batch is in: the handle is not needed any more
			}
			return curr;
		}
		@Override public int outdegree() {
			if (curr == from - 1) throw new IllegalStateException();
			return (int)(rowptr[curr - lo + 1] - rowptr[curr - lo]);
		}
		@Override public int[] successorArray() {
			final int d = outdegree();
			final int[] a = new int[d];
			System.arraycopy(succ, (int)rowptr[curr - lo], a, 0, d);
			return a;
		}
		@Override public LazyIntIterator successors() { final int[] a = successorArray(); return LazyIntIterators.wrap(a, a.length); }
		@Override public NodeIterator copy(final int upperBound) { return new BatchIterator(cloneHandle(handle), true, curr + 1, upperBound); }
	}

	@SuppressWarnings("deprecation")
	@Override protected void finalize() throws Throwable { try { close(handle); } finally { super.finalize(); } }
}
