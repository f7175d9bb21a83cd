/*
 * bvgpu_jni.c -- JNI glue between it.unimi.dsi.webgraph.gpu.GpuBVGraph and libbvgpu (include/bvgpu.h).
 * NOT COMPILED HERE (no jni.h in the build image).  Build where a JDK exists:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include bvgpu_jni.c -L../../webgraph_amd -lbvgpu -o libbvgpu_jni.so
 */
#include <jni.h>
#include <limits.h>
#include <stdlib.h>
#include "bvgpu.h"

static void throw_status(JNIEnv *env, int rc, const bvg_t *h) {
	const char *cls = rc == BVG_EARG ? "java/lang/IllegalArgumentException"
	                : rc == BVG_ESTATE ? "java/lang/IllegalStateException"
	                : rc == BVG_EUNSUPPORTED ? "java/lang/UnsupportedOperationException"
	                : rc == BVG_EIO ? "java/io/IOException"
	                : rc == BVG_ENOMEM ? "java/lang/OutOfMemoryError" : "java/lang/RuntimeException";
	(*env)->ThrowNew(env, (*env)->FindClass(env, cls), h ? bvg_last_error(h) : "bvgpu error");
}

JNIEXPORT jlong JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_open(JNIEnv *env, jclass c, jstring basename, jint device) {
	const char *b = (*env)->GetStringUTFChars(env, basename, NULL);
	bvg_t *h = NULL;
	const int rc = bvg_open(b, device, &h);
	(*env)->ReleaseStringUTFChars(env, basename, b);
	if (rc) { throw_status(env, rc, h); bvg_close(h); return 0; }
	return (jlong)(intptr_t)h;
}
JNIEXPORT jlong JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_cloneHandle(JNIEnv *env, jclass c, jlong handle) {
	bvg_t *h = NULL;
	const int rc = bvg_clone((bvg_t *)(intptr_t)handle, &h);
	if (rc) { throw_status(env, rc, h); bvg_close(h); return 0; }
	return (jlong)(intptr_t)h;
}
JNIEXPORT void JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_close(JNIEnv *env, jclass c, jlong handle) { bvg_close((bvg_t *)(intptr_t)handle); }

JNIEXPORT jlongArray JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_info(JNIEnv *env, jclass c, jlong handle) {
	bvg_info_t i;
	const int rc = bvg_info((bvg_t *)(intptr_t)handle, &i);
	if (rc) { throw_status(env, rc, (bvg_t *)(intptr_t)handle); return NULL; }
	jlong v[4] = { i.nodes, i.arcs, i.window_size, i.max_ref_count };
	jlongArray a = (*env)->NewLongArray(env, 4);
	(*env)->SetLongArrayRegion(env, a, 0, 4, v);
	return a;
}
JNIEXPORT jint JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_outdegree(JNIEnv *env, jclass c, jlong handle, jint x) {
	int32_t d = 0;
	const int rc = bvg_outdegrees((bvg_t *)(intptr_t)handle, x, x + 1, &d, BVG_OUT_HOST);
	if (rc) throw_status(env, rc, (bvg_t *)(intptr_t)handle);
	return d;
}
JNIEXPORT jintArray JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_successorArray(JNIEnv *env, jclass c, jlong handle, jint x) {
	bvg_t *h = (bvg_t *)(intptr_t)handle;
	int64_t rp[2]; uint64_t arcs = 0; int32_t node = x;
	int rc = bvg_successors_batch(h, &node, 1, rp, NULL, 0, &arcs, BVG_OUT_HOST);
	if (rc) { throw_status(env, rc, h); return NULL; }
	if (arcs > (uint64_t)INT_MAX) { throw_status(env, BVG_ENOMEM, NULL); return NULL; }
	jintArray a = (*env)->NewIntArray(env, (jsize)arcs);   /* a fresh exact-length array per call, ImmutableGraph.java:329-333 */
	if (!a) return NULL;                                    /* OutOfMemoryError is pending */
	jint *p = (*env)->GetPrimitiveArrayCritical(env, a, NULL);
	if (!p) return NULL;
	rc = bvg_successors_batch(h, &node, 1, rp, (int32_t *)p, arcs, &arcs, BVG_OUT_HOST);
	(*env)->ReleasePrimitiveArrayCritical(env, a, p, 0);
	if (rc) { throw_status(env, rc, h); return NULL; }
	return a;
}
/* ONE library call per batch: the scan runs once, its chunks cross PCIe into the handle's pinned buffers while the next
 * chunk is decoded (bvg_decode_range_view); the JVM then copies them into the arrays it owns. */
JNIEXPORT jintArray JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_decodeRange(JNIEnv *env, jclass c, jlong handle, jint from, jint to, jlongArray rowptr) {
	bvg_t *h = (bvg_t *)(intptr_t)handle;
	const int64_t *rp = NULL; const int32_t *sc = NULL;
	uint64_t arcs = 0;
	const int rc = bvg_decode_range_view(h, from, to, &rp, &sc, &arcs);
	if (rc) { throw_status(env, rc, h); return NULL; }
	if (arcs > (uint64_t)INT_MAX) { throw_status(env, BVG_ENOMEM, NULL); return NULL; } /* a Java array holds < 2^31 ints: use smaller batches */
	jintArray a = (*env)->NewIntArray(env, (jsize)arcs);
	if (!a) return NULL;                                    /* OutOfMemoryError is pending */
	(*env)->SetLongArrayRegion(env, rowptr, 0, (jsize)(to - from + 1), (const jlong *)rp);
	if ((*env)->ExceptionCheck(env)) return NULL;           /* rowptr shorter than to - from + 1: ArrayIndexOutOfBoundsException is pending, no JNI call may follow */
	(*env)->SetIntArrayRegion(env, a, 0, (jsize)arcs, (const jint *)sc);
	return a;
}
JNIEXPORT jint JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_scanChecksum(JNIEnv *env, jclass c, jlong handle, jint from, jint to, jint hash) {
	bvg_t *h = (bvg_t *)(intptr_t)handle;
	int32_t v = hash;
	const int rc = bvg_scan_checksum(h, from, to, &v, NULL);
	if (rc) throw_status(env, rc, h);
	return v;
}

JNIEXPORT jboolean JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_equalRange(JNIEnv *env, jclass c, jlong handleA, jlong handleB, jint from, jint to) {
	bvg_t *a = (bvg_t *)(intptr_t)handleA, *b = (bvg_t *)(intptr_t)handleB;
	int eq = 0;
	const int rc = bvg_equal_range(a, b, from, to, &eq);
	if (rc) throw_status(env, rc, a);
	return eq ? JNI_TRUE : JNI_FALSE;
}

static void throw_msg(JNIEnv *env, int rc, const char *msg) {
	const char *cls = rc == BVG_EARG ? "java/lang/IllegalArgumentException"
	                : rc == BVG_EUNSUPPORTED ? "java/lang/UnsupportedOperationException"
	                : rc == BVG_EIO ? "java/io/IOException"
	                : rc == BVG_ENOMEM ? "java/lang/OutOfMemoryError" : "java/lang/RuntimeException";
	(*env)->ThrowNew(env, (*env)->FindClass(env, cls), msg);
}

/* BVGraph.store for a CSR drained on the Java side (GpuBVGraph.store): bvg_store, host pointers */
JNIEXPORT void JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_storeCsr(JNIEnv *env, jclass c, jstring basename, jint device, jint n, jlongArray rowptr, jintArray succ,
                                                                          jint window, jint maxRef, jint minInterval, jint zetaK, jint flags, jint threads) {
	const char *base = (*env)->GetStringUTFChars(env, basename, NULL);
	if (!base) return;
	jlong *rp = (*env)->GetLongArrayElements(env, rowptr, NULL);
	jint *sc = rp ? (*env)->GetIntArrayElements(env, succ, NULL) : NULL;
	char err[512] = "";
	int rc = BVG_ENOMEM;
	if (rp && sc) rc = bvg_store(base, device, n, (const int64_t *)rp, (const int32_t *)sc, BVG_OUT_HOST, window, maxRef, minInterval, zetaK, (uint32_t)flags, threads, NULL, err, sizeof err);
	if (sc) (*env)->ReleaseIntArrayElements(env, succ, sc, JNI_ABORT);
	if (rp) (*env)->ReleaseLongArrayElements(env, rowptr, rp, JNI_ABORT);
	(*env)->ReleaseStringUTFChars(env, basename, base);
	if (rc && !(*env)->ExceptionCheck(env)) throw_msg(env, rc, err);
}

/* BVGraph.store when the graph is a GpuBVGraph: bvg_recompress */
JNIEXPORT void JNICALL Java_it_unimi_dsi_webgraph_gpu_GpuBVGraph_recompress(JNIEnv *env, jclass c, jlong handle, jstring basename, jint window, jint maxRef, jint minInterval,
                                                                            jint zetaK, jint flags, jint threads) {
	const char *base = (*env)->GetStringUTFChars(env, basename, NULL);
	if (!base) return;
	char err[512] = "";
	const int rc = bvg_recompress((bvg_t *)(intptr_t)handle, base, window, maxRef, minInterval, zetaK, (uint32_t)flags, threads, NULL, err, sizeof err);
	(*env)->ReleaseStringUTFChars(env, basename, base);
	if (rc) throw_msg(env, rc, err);
}
