/*
 * fmx_jni.c -- the JNI shim between findex's Scala code and libfmx.so (C ABI: include/fmx.h).
 *
 * The reference (martende/findex) has no FFI layer; its seam is the trait SuffixAlgo / SuffixWalkingAlgo
 * (src/main/scala/org/fmindex/findex.scala:9-57).  bindings/hipfm.scala holds the Scala side: `HipFMSearcher
 * extends SuffixWalkingAlgo` and the batch / regex entry points; every `@native def x0` there is
 * Java_org_fmindex_HipFM_00024_x0 here (the natives live in `object HipFM`, whose JVM class is `HipFM$`).
 *
 * Build (needs a JDK; the image this library was written in has none, so this file has never been compiled --
 * it is written against include/fmx.h as shipped and the JNI specification):
 *   cc -O2 -fPIC -shared -I"$JAVA_HOME/include" -I"$JAVA_HOME/include/linux" -Iinclude \
 *      bindings/fmx_jni.c -Lfindex_amd/lib -lfmx -o libfmx_jni.so
 *
 * Conventions: handles travel as jlong; a non-zero fmx status becomes java.lang.Exception(fmx_last_error()) --
 * the reference throws plain Exceptions for bad files too (bwtmerger.scala:153,261-262,430).  Primitive arrays
 * are COPIED in and out with Get/Set<Type>ArrayRegion around the library call: every batch entry point of libfmx
 * blocks on the GPU (stream synchronisation, device allocation, graph capture), and JNI forbids blocking inside a
 * Get/ReleasePrimitiveArrayCritical region (it stalls the collector for every thread and can deadlock when several
 * threads call in) -- critical regions are used only around plain copy loops.  Callers with large batches use the
 * Direct variants: direct ByteBuffers over fmx_host_alloc memory, no copy at all and the pipelined DMA path.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <fmx.h>

#define H(h) ((fmx_index *)(intptr_t)(h))
#define FN(name) Java_org_fmindex_HipFM_00024_##name

/* copies of primitive arrays (see the header comment); NULL when the JVM is out of memory */
static jbyte *in_bytes(JNIEnv *e, jbyteArray a, jsize n) {
  jbyte *p = malloc((size_t)(n > 0 ? n : 1));
  if (p && n > 0) (*e)->GetByteArrayRegion(e, a, 0, n, p);
  return p;
}
static jlong *in_longs(JNIEnv *e, jlongArray a, jsize n) {
  jlong *p = malloc(sizeof(jlong) * (size_t)(n > 0 ? n : 1));
  if (p && n > 0) (*e)->GetLongArrayRegion(e, a, 0, n, p);
  return p;
}

static int rethrow(JNIEnv *e, int rc) { /* returns 1 when an exception is now pending */
  if (rc == FMX_OK || rc == FMX_TRUNCATED) return 0;
  jclass ex = (*e)->FindClass(e, "java/lang/Exception");
  if (ex) (*e)->ThrowNew(e, ex, fmx_last_error());
  return 1;
}

/* ---- open / close: NaiveFMSearcher(filename, bigEndian), bwtmerger.scala:335-353 */
JNIEXPORT jlong JNICALL FN(open0)(JNIEnv *e, jobject self, jstring bwt, jstring aux, jboolean bigEndian, jint device) {
  const char *b = (*e)->GetStringUTFChars(e, bwt, 0);
  const char *a = (*e)->GetStringUTFChars(e, aux, 0);
  fmx_index *h = 0;
  int rc = (b && a) ? fmx_open(b, a, bigEndian ? 1 : 0, device, &h) : FMX_ERR_NOMEM;
  if (b) (*e)->ReleaseStringUTFChars(e, bwt, b);
  if (a) (*e)->ReleaseStringUTFChars(e, aux, a);
  rethrow(e, rc);
  return (jlong)(intptr_t)h;
}

/* NaiveBWTSearcher(bwt, bucketStarts, rk0), findex.scala:459-506 */
JNIEXPORT jlong JNICALL FN(openBlock0)(JNIEnv *e, jobject self, jbyteArray bwt, jlongArray bucketStarts, jlong rk0,
                                       jint device) {
  if ((*e)->GetArrayLength(e, bucketStarts) != 256) { rethrow(e, FMX_ERR_ARG); return 0; }
  jsize n = (*e)->GetArrayLength(e, bwt);
  jbyte *b = in_bytes(e, bwt, n);
  jlong *bs = in_longs(e, bucketStarts, 256);
  fmx_index *h = 0;
  int rc = (b && bs) ? fmx_open_block((const uint8_t *)b, (uint64_t)n, (const int64_t *)bs, (uint64_t)rk0, device, &h)
                     : FMX_ERR_NOMEM;
  free(bs);
  free(b);
  rethrow(e, rc);
  return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL FN(close0)(JNIEnv *e, jobject self, jlong h) { fmx_close(H(h)); }

/* ---- scalars: SuffixAlgo.n / cf, findex.scala:10-12 */
JNIEXPORT jlong JNICALL FN(n0)(JNIEnv *e, jobject self, jlong h) {
  uint64_t v = 0;
  rethrow(e, fmx_n(H(h), &v));
  return (jlong)v;
}
JNIEXPORT jlong JNICALL FN(eof0)(JNIEnv *e, jobject self, jlong h) {
  uint64_t v = 0;
  rethrow(e, fmx_eof(H(h), &v));
  return (jlong)v;
}
JNIEXPORT jlong JNICALL FN(cf0)(JNIEnv *e, jobject self, jlong h, jint c) {
  uint64_t v = 0;
  rethrow(e, fmx_cf(H(h), c, &v));
  return (jlong)v;
}

/* ---- occ(c, i) for k pairs: out[q] = occ(c[q], i[q]) */
JNIEXPORT void JNICALL FN(occBatch0)(JNIEnv *e, jobject self, jlong h, jbyteArray c, jlongArray i, jlongArray out) {
  jsize k = (*e)->GetArrayLength(e, c);
  if ((*e)->GetArrayLength(e, i) != k || (*e)->GetArrayLength(e, out) < k) { rethrow(e, FMX_ERR_ARG); return; }
  jbyte *pc = in_bytes(e, c, k);
  jlong *pi = in_longs(e, i, k);
  jlong *po = malloc(sizeof(jlong) * (size_t)(k > 0 ? k : 1));
  int rc = (pc && pi && po) ? fmx_occ_batch(H(h), (const uint8_t *)pc, (const int64_t *)pi, (uint64_t *)po, (size_t)k)
                            : FMX_ERR_NOMEM;
  if (rc == FMX_OK && k > 0) (*e)->SetLongArrayRegion(e, out, 0, k, po);
  free(po);
  free(pi);
  free(pc);
  rethrow(e, rc);
}

/* ---- search for k patterns: pat = all pattern bytes, off = k+1 offsets, out = long[2k] (sp[0..k) then ep[0..k)) */
JNIEXPORT void JNICALL FN(searchBatch0)(JNIEnv *e, jobject self, jlong h, jbyteArray pat, jlongArray off, jlongArray out) {
  jsize k = (*e)->GetArrayLength(e, off) - 1;
  if (k < 0 || (*e)->GetArrayLength(e, out) < 2 * k) { rethrow(e, FMX_ERR_ARG); return; }
  jbyte *p = in_bytes(e, pat, (*e)->GetArrayLength(e, pat));
  jlong *o = in_longs(e, off, k + 1);
  jlong *r = malloc(sizeof(jlong) * (size_t)(k > 0 ? 2 * k : 1));
  /* (what becomes of these pairs is SuffixAlgo.search's Option: a miss is None whatever its values -- FMX_SEARCH_MISS_NONE
   * spares the device the walk to the reference loop's values at the failing step) */
  fmx_search_opts opts = {0, FMX_SEARCH_MISS_NONE, 0};
  int rc = (p && o && r) ? fmx_search_batch_ex(H(h), (const uint8_t *)p, (const uint64_t *)o, (uint64_t *)r,
                                               (uint64_t *)r + k, (size_t)k, &opts)
                         : FMX_ERR_NOMEM;
  if (rc == FMX_OK && k > 0) (*e)->SetLongArrayRegion(e, out, 0, 2 * k, r);
  free(r);
  free(o);
  free(p);
  rethrow(e, rc);
}

/* The same over direct ByteBuffers (little-endian longs): with buffers from hostAlloc0 the library moves the batch
 * by DMA, pipelined against the search (fmx_search_batch, include/fmx.h).  out = 16k bytes: sp[0..k) then ep[0..k). */
JNIEXPORT void JNICALL FN(searchBatchDirect0)(JNIEnv *e, jobject self, jlong h, jobject pat, jobject off, jobject out,
                                              jlong k) {
  uint8_t *p = (*e)->GetDirectBufferAddress(e, pat);
  uint64_t *o = (*e)->GetDirectBufferAddress(e, off);
  uint64_t *r = (*e)->GetDirectBufferAddress(e, out);
  if (!o || !r || k < 0 || (*e)->GetDirectBufferCapacity(e, off) < 8 * (k + 1) ||
      (*e)->GetDirectBufferCapacity(e, out) < 16 * k) {
    rethrow(e, FMX_ERR_ARG);
    return;
  }
  rethrow(e, fmx_search_batch(H(h), p, o, r, r + k, (size_t)k));
}

/* ---- the lean form (fmx_search_batch_ex): k = pat.length / len equal-length patterns, no offsets; out = long[2k] */
JNIEXPORT void JNICALL FN(searchBatchFixed0)(JNIEnv *e, jobject self, jlong h, jbyteArray pat, jint len, jlongArray out) {
  jsize nb = (*e)->GetArrayLength(e, pat);
  if (len <= 0 || nb % len != 0 || (*e)->GetArrayLength(e, out) < 2 * (nb / len)) { rethrow(e, FMX_ERR_ARG); return; }
  jsize k = nb / len;
  jlong *r = malloc(sizeof(jlong) * (size_t)(k > 0 ? 2 * k : 1));
  if (!r) { rethrow(e, FMX_ERR_NOMEM); return; }      /* before the pattern bytes are copied out of the JVM */
  jbyte *p = in_bytes(e, pat, nb);
  fmx_search_opts opts = {(uint32_t)len, FMX_SEARCH_MISS_NONE, 0};
  int rc = (p && r) ? fmx_search_batch_ex(H(h), (const uint8_t *)p, 0, (uint64_t *)r, (uint64_t *)r + k, (size_t)k, &opts) : FMX_ERR_NOMEM;
  if (rc == FMX_OK && k > 0) (*e)->SetLongArrayRegion(e, out, 0, 2 * k, r);
  free(r);
  free(p);
  rethrow(e, rc);
}

/* the same over direct ByteBuffers (fmx_host_alloc memory) with the intervals back in the 8-byte form: `out` holds
 * fmx_packed_words(k, escapeCap) longs; HipFM.unpack decodes them (word q = sp | width << 40, escape list behind) */
JNIEXPORT void JNICALL FN(searchBatchPackedDirect0)(JNIEnv *e, jobject self, jlong h, jobject pat, jint len, jobject out, jlong k,
                                                    jlong escapeCap) {
  uint8_t *p = (*e)->GetDirectBufferAddress(e, pat);
  uint64_t *r = (*e)->GetDirectBufferAddress(e, out);
  /* escapeCap <= k: more escape entries than patterns are never needed -- and it keeps 8 * (k + 1 + 2 * escapeCap) from
   * wrapping (k * len fits the pattern buffer's jlong capacity, so k < 2^63 / len; 3 k + 1 words then fit 64 bits for any
   * len >= 1 only below 2^59: checked) */
  if (!p || !r || k < 0 || len <= 0 || escapeCap < 0 || escapeCap > k || k > ((jlong)1 << 58) ||
      (*e)->GetDirectBufferCapacity(e, pat) / (jlong)len < k ||
      (*e)->GetDirectBufferCapacity(e, out) / 8 < (jlong)fmx_packed_words((size_t)k, (size_t)escapeCap)) {
    rethrow(e, FMX_ERR_ARG);
    return;
  }
  fmx_search_opts opts = {(uint32_t)len, 1, (uint64_t)escapeCap};
  rethrow(e, fmx_search_batch_ex(H(h), p, 0, r, 0, (size_t)k, &opts));
}

/* ---- getPrevRange for k (sp, ep, c) triples: out = long[2k] (sp1[0..k) then ep1[0..k)) */
JNIEXPORT void JNICALL FN(prevRangeBatch0)(JNIEnv *e, jobject self, jlong h, jlongArray sp, jlongArray ep, jbyteArray c,
                                           jlongArray out) {
  jsize k = (*e)->GetArrayLength(e, c);
  if ((*e)->GetArrayLength(e, sp) != k || (*e)->GetArrayLength(e, ep) != k || (*e)->GetArrayLength(e, out) < 2 * k) {
    rethrow(e, FMX_ERR_ARG);
    return;
  }
  jlong *ps = in_longs(e, sp, k);
  jlong *pe = in_longs(e, ep, k);
  jbyte *pc = in_bytes(e, c, k);
  jlong *po = malloc(sizeof(jlong) * (size_t)(k > 0 ? 2 * k : 1));
  int rc = (ps && pe && pc && po)
               ? fmx_prev_range_batch(H(h), (const uint64_t *)ps, (const uint64_t *)pe, (const uint8_t *)pc,
                                      (uint64_t *)po, (uint64_t *)po + k, (size_t)k)
               : FMX_ERR_NOMEM;
  if (rc == FMX_OK && k > 0) (*e)->SetLongArrayRegion(e, out, 0, 2 * k, po);
  free(po);
  free(pc);
  free(pe);
  free(ps);
  rethrow(e, rc);
}

/* ---- getIntervalPrevRange: out = long[2 * (cend - cstart + 1)] as (sp, ep) pairs in the reference's descending-c
 * order; returns the number of pairs */
JNIEXPORT jint JNICALL FN(intervalPrevRange0)(JNIEnv *e, jobject self, jlong h, jlong sp, jlong ep, jint cstart,
                                              jint cend, jlongArray out) {
  jsize cap = (*e)->GetArrayLength(e, out) / 2;
  jsize want = cend >= cstart ? cend - cstart + 1 : 0;
  if (cap < want) { rethrow(e, FMX_ERR_ARG); return 0; }
  uint64_t *osp = malloc(sizeof(uint64_t) * (size_t)(want ? want : 1));
  uint64_t *oep = malloc(sizeof(uint64_t) * (size_t)(want ? want : 1));
  size_t got = 0;
  int rc = (osp && oep) ? fmx_interval_prev_range(H(h), (uint64_t)sp, (uint64_t)ep, cstart, cend, osp, oep, 0, &got)
                        : FMX_ERR_NOMEM;
  if (rc == FMX_OK && got) {
    jlong *po = (*e)->GetPrimitiveArrayCritical(e, out, 0);
    if (po) {
      for (size_t j = 0; j < got; j++) { po[2 * j] = (jlong)osp[j]; po[2 * j + 1] = (jlong)oep[j]; }
      (*e)->ReleasePrimitiveArrayCritical(e, out, po, 0);
    } else {
      rc = FMX_ERR_NOMEM;
    }
  }
  free(osp);
  free(oep);
  rethrow(e, rc);
  return (jint)got;
}

/* ---- walkers: nextSubstr / prevSubstr (bwtmerger.scala:394-419) as byte arrays */
static jbyteArray extract(JNIEnv *e, jlong h, jlong row, jint len, int direction) {
  uint8_t *buf = malloc((size_t)(len > 0 ? len : 1));
  uint32_t got = 0;
  int rc = buf ? fmx_extract(H(h), (uint64_t)row, (uint32_t)(len > 0 ? len : 0), direction, buf, &got) : FMX_ERR_NOMEM;
  jbyteArray r = 0;
  if (!rethrow(e, rc)) {
    r = (*e)->NewByteArray(e, (jsize)got);
    if (r && got) (*e)->SetByteArrayRegion(e, r, 0, (jsize)got, (const jbyte *)buf);
  }
  free(buf);
  return r;
}
JNIEXPORT jbyteArray JNICALL FN(nextSubstr0)(JNIEnv *e, jobject self, jlong h, jlong sp, jint len) {
  return extract(e, h, sp, len, 1);
}
JNIEXPORT jbyteArray JNICALL FN(prevSubstr0)(JNIEnv *e, jobject self, jlong h, jlong sp, jint len) {
  return extract(e, h, sp, len, -1);
}
/* nextSubstr for k rows at once (rendering a result list): out = byte[k * len], outLen = int[k] */
JNIEXPORT void JNICALL FN(nextSubstrBatch0)(JNIEnv *e, jobject self, jlong h, jlongArray rows, jint len, jbyteArray out,
                                            jintArray outLen) {
  jsize k = (*e)->GetArrayLength(e, rows);
  if (len < 0 || (*e)->GetArrayLength(e, outLen) < k || (jlong)(*e)->GetArrayLength(e, out) < (jlong)k * len) {
    rethrow(e, FMX_ERR_ARG);
    return;
  }
  jlong *pr = in_longs(e, rows, k);
  const size_t nbytes = (size_t)k * (size_t)len;
  jbyte *po = malloc(nbytes ? nbytes : 1);
  jint *pl = malloc(sizeof(jint) * (size_t)(k > 0 ? k : 1));
  int rc = (pr && po && pl) ? fmx_next_substr_batch(H(h), (const uint64_t *)pr, (size_t)k, (uint32_t)len, (uint8_t *)po,
                                                    (uint32_t *)pl)
                            : FMX_ERR_NOMEM;
  if (rc == FMX_OK && k > 0) {
    if (nbytes) (*e)->SetByteArrayRegion(e, out, 0, (jsize)nbytes, po);
    (*e)->SetIntArrayRegion(e, outLen, 0, k, pl);
  }
  free(pl);
  free(po);
  free(pr);
  rethrow(e, rc);
}

/* ---- FMCreator.create (bwtmerger.scala:424-533): write the reference's .fm file */
JNIEXPORT void JNICALL FN(writeFm0)(JNIEnv *e, jobject self, jlong h, jstring path) {
  const char *p = (*e)->GetStringUTFChars(e, path, 0);
  int rc = p ? fmx_write_fm(H(h), p) : FMX_ERR_NOMEM;
  if (p) (*e)->ReleaseStringUTFChars(e, path, p);
  rethrow(e, rc);
}

/* The same into direct ByteBuffers: out = cap fmx_result records of 24 bytes (u32 regex, u32 len, u64 sp, u64 ep,
 * native byte order), perRegex = k u32 counts or null.  With buffers from hostAlloc0 the device writes the grouped
 * results and the counts into them itself, behind the search and before the call's one synchronisation
 * (k_res_export): nothing is copied or converted on the host -- the form a serving loop uses on a resident batch. */
JNIEXPORT jlong JNICALL FN(regexBatchMatchDirect0)(JNIEnv *e, jobject self, jlong h, jlong batch, jintArray limits,
                                                   jlong maxFrontier, jobject out, jobject perRegex, jintArray status) {
  jint lim4[4] = {0, 0, 1024, 1000};
  (*e)->GetIntArrayRegion(e, limits, 0, 4, lim4);
  fmx_limits lim;
  lim.max_steps = (uint32_t)lim4[0];
  lim.mode = (uint32_t)lim4[1];
  lim.max_frontier = (uint64_t)maxFrontier;
  lim.max_branching = (uint32_t)lim4[2];
  lim.max_iterations = (uint32_t)lim4[3];
  fmx_result *res = (*e)->GetDirectBufferAddress(e, out);
  uint32_t *per = perRegex ? (*e)->GetDirectBufferAddress(e, perRegex) : 0;
  if (!res || (perRegex && !per)) {
    rethrow(e, FMX_ERR_ARG);
    return 0;
  }
  size_t cap = (size_t)((*e)->GetDirectBufferCapacity(e, out) / (jlong)sizeof(fmx_result)), got = 0;
  uint64_t k_batch = 0;
  if (rethrow(e, fmx_regex_batch_info((const fmx_regex_batch *)(intptr_t)batch, &k_batch, 0, 0, 0))) return 0;
  if (per && (uint64_t)(*e)->GetDirectBufferCapacity(e, perRegex) < 4 * k_batch) { rethrow(e, FMX_ERR_ARG); return 0; }
  int rc = fmx_regex_batch_match(H(h), (fmx_regex_batch *)(intptr_t)batch, &lim, res, cap, &got, per);
  if ((rc == FMX_OK || rc == FMX_TRUNCATED) && status) {
    jint st = rc == FMX_TRUNCATED ? 1 : 0;
    (*e)->SetIntArrayRegion(e, status, 0, 1, &st);
  }
  rethrow(e, rc);
  return (jlong)got;
}

/* ---- page-locked batch buffers as direct ByteBuffers */
JNIEXPORT jobject JNICALL FN(hostAlloc0)(JNIEnv *e, jobject self, jlong bytes) {
  void *p = 0;
  if (rethrow(e, fmx_host_alloc((size_t)bytes, &p))) return 0;
  return (*e)->NewDirectByteBuffer(e, p, bytes);
}
JNIEXPORT void JNICALL FN(hostFree0)(JNIEnv *e, jobject self, jobject buf) {
  void *p = (*e)->GetDirectBufferAddress(e, buf);
  if (p) rethrow(e, fmx_host_free(p));
}

/* ---- regex: REParser.re2post + ReTree.apply (re2/re2.scala:50-185, re2/retree.scala:156-370).
 * The string goes over as Latin-1 bytes (the reference's Chars are bytes). */
JNIEXPORT jlong JNICALL FN(regexCompile0)(JNIEnv *e, jobject self, jbyteArray latin1, jboolean lineOnly) {
  jsize n = (*e)->GetArrayLength(e, latin1);
  char *s = malloc((size_t)n + 1);
  fmx_regex *r = 0;
  int rc = FMX_ERR_NOMEM;
  if (s) {
    (*e)->GetByteArrayRegion(e, latin1, 0, n, (jbyte *)s);
    s[n] = 0;
    rc = fmx_regex_compile(s, lineOnly ? 1 : 0, &r);
    free(s);
  }
  rethrow(e, rc);       /* FMX_ERR_SYNTAX = "re2post syntax", FMX_ERR_MATCH = scala.MatchError */
  return (jlong)(intptr_t)r;
}
JNIEXPORT void JNICALL FN(regexFree0)(JNIEnv *e, jobject self, jlong r) { fmx_regex_free((fmx_regex *)(intptr_t)r); }

/* k regexes in one call, compiled on all host cores inside the library (fmx_regex_compile_batch): `packed` = the k
 * Latin-1 strings, each followed by a 0 byte; handles[i] = regex i's handle or 0, status[i] = 0 / FMX_ERR_SYNTAX (7) /
 * FMX_ERR_MATCH (8) as regexCompile0 would throw for it.  No exception for a regex that does not compile. */
JNIEXPORT void JNICALL FN(regexCompileBatch0)(JNIEnv *e, jobject self, jbyteArray packed, jint k, jboolean lineOnly,
                                              jlongArray handles, jintArray status) {
  jsize n = (*e)->GetArrayLength(e, packed);
  if (k < 0 || (*e)->GetArrayLength(e, handles) < k || (*e)->GetArrayLength(e, status) < k) { rethrow(e, FMX_ERR_ARG); return; }
  jbyte *buf = in_bytes(e, packed, n);
  const char **ptr = malloc(sizeof(char *) * (size_t)(k ? k : 1));
  fmx_regex **out = malloc(sizeof(fmx_regex *) * (size_t)(k ? k : 1));
  jint *st = malloc(sizeof(jint) * (size_t)(k ? k : 1));
  jlong *hl = malloc(sizeof(jlong) * (size_t)(k ? k : 1));
  int rc = FMX_ERR_NOMEM;
  if (buf && ptr && out && st && hl) {
    jsize at = 0, found = 0;
    for (; found < k && at < n; found++) {           /* the start of each 0-terminated string */
      ptr[found] = (const char *)buf + at;
      while (at < n && buf[at] != 0) at++;
      if (at >= n) break;                              /* unterminated */
      at++;
    }
    rc = found == k ? fmx_regex_compile_batch(ptr, (size_t)k, lineOnly ? 1 : 0, out, (int *)st) : FMX_ERR_ARG;
    if (rc == FMX_OK) {
      for (jint j = 0; j < k; j++) hl[j] = (jlong)(intptr_t)out[j];
      (*e)->SetLongArrayRegion(e, handles, 0, k, hl);
      (*e)->SetIntArrayRegion(e, status, 0, k, st);
    }
  }
  free(hl); free(st); free(out); free(ptr); free(buf);
  rethrow(e, rc);
}
JNIEXPORT void JNICALL FN(regexFreeBatch0)(JNIEnv *e, jobject self, jlongArray handles) {
  jsize k = (*e)->GetArrayLength(e, handles);
  jlong *hl = in_longs(e, handles, k);
  if (hl) for (jsize j = 0; j < k; j++) fmx_regex_free((fmx_regex *)(intptr_t)hl[j]);
  free(hl);
}

/* ---- BWTMerger2.calcGaps' rank loop (bwtmerger.scala:981-1023) on the host-side dictionary */
JNIEXPORT jlong JNICALL FN(occHost0)(JNIEnv *e, jobject self, jlong h, jint c, jlong i) {
  uint64_t v = 0;
  rethrow(e, fmx_occ_host(H(h), c, (int64_t)i, &v));
  return (jlong)v;
}
JNIEXPORT jint JNICALL FN(calcGapsChain0)(JNIEnv *e, jobject self, jlong h, jbyteArray text, jint from, jlong rank0,
                                          jint lastChar, jlong rklst, jlongArray ranks) {
  jsize n = (*e)->GetArrayLength(e, text);
  if (from < 0 || from > n || (*e)->GetArrayLength(e, ranks) < n - from) { rethrow(e, FMX_ERR_ARG); return 0; }
  jsize k = n - from;
  jbyte *t = malloc((size_t)(k > 0 ? k : 1));
  jlong *r = malloc(sizeof(jlong) * (size_t)(k > 0 ? k : 1));
  size_t done = 0;
  int rc = FMX_ERR_NOMEM;
  if (t && r) {
    if (k > 0) (*e)->GetByteArrayRegion(e, text, from, k, t);
    rc = fmx_calc_gaps_chain(H(h), (const uint8_t *)t, (size_t)k, (uint64_t)rank0, lastChar, (uint64_t)rklst, (uint64_t *)r, &done);
    if (rc == FMX_OK && k > 0) (*e)->SetLongArrayRegion(e, ranks, 0, (jsize)(done < (size_t)k ? done + 1 : done), r);
  }
  free(r);
  free(t);
  rethrow(e, rc);
  return (jint)done;
}

/* fmx_prepare: build the k-mer jump table (what & 1) / the select directory (what & 2) / the literal search's row tables (what & 4) /
 * the regex frontier's row table (what & 8) now, not at the threshold or at first use */
JNIEXPORT void JNICALL FN(prepare0)(JNIEnv *e, jobject self, jlong h, jint what) { rethrow(e, fmx_prepare(H(h), (unsigned)what)); }

/* fmx_prepare_ex: the same under a budget of device bytes for all derived tables of the handle (0: the handle's own policy) */
JNIEXPORT void JNICALL FN(prepareEx0)(JNIEnv *e, jobject self, jlong h, jint what, jlong budgetBytes) {
  rethrow(e, budgetBytes < 0 ? FMX_ERR_ARG : fmx_prepare_ex((fmx_index *)H(h), (unsigned)what, (uint64_t)budgetBytes));
}

/* fmx_index_config_set: one handle's own table policy ("ktab", "jump", "jump_pairs", "jump_chars", "tables_after", "table_budget") */
JNIEXPORT void JNICALL FN(indexConfigSet0)(JNIEnv *e, jobject self, jlong h, jstring key, jstring value) {
  const char *k = (*e)->GetStringUTFChars(e, key, 0);
  const char *v = (*e)->GetStringUTFChars(e, value, 0);
  int rc = (k && v) ? fmx_index_config_set((fmx_index *)H(h), k, v) : FMX_ERR_NOMEM;
  if (v) (*e)->ReleaseStringUTFChars(e, value, v);
  if (k) (*e)->ReleaseStringUTFChars(e, key, k);
  rethrow(e, rc);
}

/* fmx_drop_tables: free derived tables again (what & 1: the k-mer table, what & 4: J and R3, what & 8: the frontier's) */
JNIEXPORT void JNICALL FN(dropTables0)(JNIEnv *e, jobject self, jlong h, jint what) { rethrow(e, fmx_drop_tables(H(h), (unsigned)what)); }

/* fmx_config_set: process-wide settings ("layout", "checkpoints", "ktab", "jump", "tables_after", "pipeline", "validate", "threads") */
JNIEXPORT void JNICALL FN(configSet0)(JNIEnv *e, jobject self, jstring key, jstring value) {
  const char *k = (*e)->GetStringUTFChars(e, key, 0);
  const char *v = (*e)->GetStringUTFChars(e, value, 0);
  int rc = (k && v) ? fmx_config_set(k, v) : FMX_ERR_NOMEM;
  if (v) (*e)->ReleaseStringUTFChars(e, value, v);
  if (k) (*e)->ReleaseStringUTFChars(e, key, k);
  rethrow(e, rc);
}

JNIEXPORT jlong JNICALL FN(regexBatchCreate0)(JNIEnv *e, jobject self, jlong h, jlongArray regexes) {
  jsize k = (*e)->GetArrayLength(e, regexes);
  fmx_regex **arr = malloc(sizeof(fmx_regex *) * (size_t)(k ? k : 1));
  fmx_regex_batch *b = 0;
  int rc = FMX_ERR_NOMEM;
  jlong *pr = in_longs(e, regexes, k);
  if (arr && pr) {
    for (jsize j = 0; j < k; j++) arr[j] = (fmx_regex *)(intptr_t)pr[j];
    rc = fmx_regex_batch_create(H(h), arr, (size_t)k, &b);
  }
  free(pr);
  free(arr);
  rethrow(e, rc);
  return (jlong)(intptr_t)b;
}
JNIEXPORT void JNICALL FN(regexBatchFree0)(JNIEnv *e, jobject self, jlong b) {
  fmx_regex_batch_free((fmx_regex_batch *)(intptr_t)b);
}

/* ReTree.matchSA over a resident batch.  limits = {max_steps, mode, max_branching, max_iterations} (ints) with
 * maxFrontier beside them; out = long[3 * cap]: per result (regex << 32 | len), sp, ep; perRegex = int[k] or null.
 * Returns the number of results; status[0] = 1 when the search was cut at max_steps (FMX_TRUNCATED). */
JNIEXPORT jlong JNICALL FN(regexBatchMatch0)(JNIEnv *e, jobject self, jlong h, jlong batch, jintArray limits,
                                             jlong maxFrontier, jlongArray out, jintArray perRegex, jintArray status) {
  jint lim4[4] = {0, 0, 1024, 1000};
  (*e)->GetIntArrayRegion(e, limits, 0, 4, lim4);
  fmx_limits lim;
  lim.max_steps = (uint32_t)lim4[0];
  lim.mode = (uint32_t)lim4[1];
  lim.max_frontier = (uint64_t)maxFrontier;
  lim.max_branching = (uint32_t)lim4[2];
  lim.max_iterations = (uint32_t)lim4[3];
  size_t cap = (size_t)(*e)->GetArrayLength(e, out) / 3, got = 0;
  uint64_t k_batch = 0;
  if (rethrow(e, fmx_regex_batch_info((const fmx_regex_batch *)(intptr_t)batch, &k_batch, 0, 0, 0))) return 0;
  jsize kper = perRegex ? (*e)->GetArrayLength(e, perRegex) : 0;
  if (perRegex && (uint64_t)kper < k_batch) {      /* the library writes one count per regex of the batch */
    jclass ex = (*e)->FindClass(e, "java/lang/IllegalArgumentException");
    if (ex) (*e)->ThrowNew(e, ex, "perRegex is shorter than the batch");
    return 0;
  }
  kper = perRegex ? (jsize)k_batch : 0;
  fmx_result *res = malloc(sizeof(fmx_result) * (cap ? cap : 1));
  uint32_t *per = 0;
  if (kper) per = malloc(sizeof(uint32_t) * (size_t)kper);
  int rc = (res && (!kper || per))
               ? fmx_regex_batch_match(H(h), (fmx_regex_batch *)(intptr_t)batch, &lim, res, cap, &got, per)
               : FMX_ERR_NOMEM;
  if (rc == FMX_OK || rc == FMX_TRUNCATED) {
    jlong *po = (*e)->GetPrimitiveArrayCritical(e, out, 0);
    if (po) {
      for (size_t j = 0; j < got; j++) {
        po[3 * j] = ((jlong)res[j].regex << 32) | (jlong)res[j].len;
        po[3 * j + 1] = (jlong)res[j].sp;
        po[3 * j + 2] = (jlong)res[j].ep;
      }
      (*e)->ReleasePrimitiveArrayCritical(e, out, po, 0);
    } else {
      rc = FMX_ERR_NOMEM;
    }
    if (kper) (*e)->SetIntArrayRegion(e, perRegex, 0, kper, (const jint *)per);
    if (status) {
      jint st = rc == FMX_TRUNCATED ? 1 : 0;
      (*e)->SetIntArrayRegion(e, status, 0, 1, &st);
    }
  }
  free(res);
  free(per);
  rethrow(e, rc);
  return (jlong)got;
}

/* ---- statistics: {rank_queries, backward_steps, launches, index_bytes, search_requests, frontier_requests} and
 * last_kernel_ms through the double array */
JNIEXPORT void JNICALL FN(stats0)(JNIEnv *e, jobject self, jlong h, jlongArray counters, jdoubleArray ms) {
  fmx_stats_t s;
  if (rethrow(e, fmx_stats(H(h), &s))) return;
  jlong c[6] = {(jlong)s.rank_queries, (jlong)s.backward_steps, (jlong)s.launches,
                (jlong)s.index_bytes,  (jlong)s.search_requests, (jlong)s.frontier_requests};
  jdouble d[2] = {s.last_kernel_ms, s.build_ms};
  (*e)->SetLongArrayRegion(e, counters, 0, 6, c);
  (*e)->SetDoubleArrayRegion(e, ms, 0, 2, d);
}
