// hipfm.scala -- the reference-side adapter: findex's search engines on libfmx.so (MI355X).
//
// Drop this file into src/main/scala/org/fmindex/ next to findex.scala and load libfmx_jni.so (bindings/fmx_jni.c)
// and libfmx.so.  It adds, and changes nothing else:
//   class HipFMSearcher(filename, bigEndian)  extends SuffixWalkingAlgo   -- drop-in for NaiveFMSearcher
//                                                                            (bwtmerger.scala:335-421)
//   class HipBWTSearcher(bwt, bucketStarts, rk0) extends SuffixAlgo       -- drop-in for NaiveBWTSearcher
//                                                                            (findex.scala:459-506)
//   object HipRegex                                                       -- ReTree.matchSA (re2/retree.scala:570-653)
//                                                                            for one regex or a batch, on the device
// Every engine of the reference takes `sa: SuffixWalkingAlgo` and touches only sa.n, sa.getPrevRange and
// sa.nextSubstr (re2/retree.scala:576,633; re2/re2.scala:13,461,473,530; dfa.scala:234,248,263), so
// `ReTree(post).matchSA(new HipFMSearcher(f))` already runs unchanged -- one kernel launch per getPrevRange, correct
// but slow.  The fast forms are the batch methods below and HipRegex, which hand whole batches to the device.
//
// Positions are Int like the trait (the reference is 32-bit, findex.scala:10-13): valid for n < 2^31.  The Long
// variants (searchBatchLong ...) serve larger indexes.  Written for Scala 2.10 like the reference
// (project/Build.scala:12); never compiled in the image this library was built in (no JVM there).
package org.fmindex

import java.nio.{ByteBuffer, ByteOrder}
import java.util.concurrent.atomic.{AtomicBoolean, AtomicLong}
import org.fmindex.re2.SAResult

object HipFM {
  System.loadLibrary("fmx_jni")
  @native def open0(bwt: String, aux: String, bigEndian: Boolean, device: Int): Long
  @native def openBlock0(bwt: Array[Byte], bucketStarts: Array[Long], rk0: Long, device: Int): Long
  @native def close0(h: Long): Unit
  @native def n0(h: Long): Long
  @native def eof0(h: Long): Long
  @native def cf0(h: Long, c: Int): Long
  @native def occBatch0(h: Long, c: Array[Byte], i: Array[Long], out: Array[Long]): Unit
  @native def searchBatch0(h: Long, pat: Array[Byte], off: Array[Long], out: Array[Long]): Unit
  @native def searchBatchDirect0(h: Long, pat: ByteBuffer, off: ByteBuffer, out: ByteBuffer, k: Long): Unit
  @native def prevRangeBatch0(h: Long, sp: Array[Long], ep: Array[Long], c: Array[Byte], out: Array[Long]): Unit
  @native def intervalPrevRange0(h: Long, sp: Long, ep: Long, cstart: Int, cend: Int, out: Array[Long]): Int
  @native def nextSubstr0(h: Long, sp: Long, len: Int): Array[Byte]
  @native def prevSubstr0(h: Long, sp: Long, len: Int): Array[Byte]
  @native def nextSubstrBatch0(h: Long, rows: Array[Long], len: Int, out: Array[Byte], outLen: Array[Int]): Unit
  @native def writeFm0(h: Long, path: String): Unit
  @native def hostAlloc0(bytes: Long): ByteBuffer
  @native def hostFree0(buf: ByteBuffer): Unit
  @native def regexCompile0(latin1: Array[Byte], lineOnly: Boolean): Long
  @native def regexFree0(r: Long): Unit
  @native def regexCompileBatch0(packed: Array[Byte], k: Int, lineOnly: Boolean, handles: Array[Long], status: Array[Int]): Unit
  @native def regexFreeBatch0(handles: Array[Long]): Unit
  @native def searchBatchFixed0(h: Long, pat: Array[Byte], len: Int, out: Array[Long]): Unit
  @native def searchBatchPackedDirect0(h: Long, pat: ByteBuffer, len: Int, out: ByteBuffer, k: Long, escapeCap: Long): Unit
  @native def dropTables0(h: Long, what: Int): Unit
  @native def prepare0(h: Long, what: Int): Unit
  @native def prepareEx0(h: Long, what: Int, budgetBytes: Long): Unit
  @native def indexConfigSet0(h: Long, key: String, value: String): Unit
  @native def configSet0(key: String, value: String): Unit
  @native def occHost0(h: Long, c: Int, i: Long): Long
  @native def calcGapsChain0(h: Long, text: Array[Byte], from: Int, rank0: Long, lastChar: Int, rklst: Long, ranks: Array[Long]): Int
  @native def regexBatchCreate0(h: Long, regexes: Array[Long]): Long
  @native def regexBatchFree0(b: Long): Unit
  @native def regexBatchMatch0(h: Long, batch: Long, limits: Array[Int], maxFrontier: Long, out: Array[Long],
                               perRegex: Array[Int], status: Array[Int]): Long
  @native def regexBatchMatchDirect0(h: Long, batch: Long, limits: Array[Int], maxFrontier: Long, out: ByteBuffer,
                                     perRegex: ByteBuffer, status: Array[Int]): Long
  @native def stats0(h: Long, counters: Array[Long], ms: Array[Double]): Unit

  val MATCH_FRONTIER = 0      // every match, breadth of the whole batch at once (the throughput path)
  val MATCH_REFERENCE = 1     // ReTree._matchSA's own queue order and limits (re2/retree.scala:618-653)

  def latin1(s: String): Array[Byte] = s.map(_.toByte).toArray
  def fromLatin1(b: Array[Byte]): String = new String(b.map(x => (x & 0xff).toChar))
}

/** The SuffixAlgo part shared by both searchers: everything is answered by the device through the handle `h`. */
trait HipSuffixAlgo extends SuffixAlgo {
  import HipFM._
  /** The native handle, 0 once closed.  close() and the finalizer both go through `release`, which takes the handle out of
    * the box exactly once: a caller that closes and is later finalized (or closes twice) reaches fmx_close once, not twice
    * -- round 4 passed an immutable `h` to close0 from both and freed the index twice inside the JVM. */
  protected def hbox: AtomicLong
  protected def h: Long = {
    val v = hbox.get
    if (v == 0L) throw new IllegalStateException("HipFM: the index handle is closed")
    v
  }
  private def release(): Unit = { val v = hbox.getAndSet(0L); if (v != 0L) close0(v) }
  lazy val n: Int = n0(h).toInt
  def nLong: Long = n0(h)
  def cf(c: Int): Int = cf0(h, c).toInt
  def occ(c: Int, i: Int): Int = occBatch(Array(c.toByte), Array(i.toLong))(0).toInt

  /** occ for many (c, i) pairs in one launch. */
  def occBatch(c: Array[Byte], i: Array[Long]): Array[Long] = {
    val out = new Array[Long](c.length); occBatch0(h, c, i, out); out
  }

  override def search(in: Array[Byte]): Option[(Int, Int)] = searchBatch(Array(in))(0)

  /** SuffixAlgo.search (findex.scala:15-31) for a batch: one launch.  A miss comes back as sp == ep and maps to
    * None like the reference's `if (sp < ep) Some(..) else None` (:30).  Bytes >= 0x80 are legal here (the
    * reference throws ArrayIndexOutOfBounds on its signed Byte index, :21,26). */
  def searchBatch(pats: Array[Array[Byte]]): Array[Option[(Int, Int)]] =
    searchBatchLong(pats).map { case (sp, ep) => if (sp < ep) Some((sp.toInt, ep.toInt)) else None }

  // (sp, ep) per pattern; a pattern that does not occur has sp >= ep and nothing else is promised of its pair (the library is
  // asked with FMX_SEARCH_MISS_NONE: SuffixAlgo.search maps it to None either way, findex.scala:30)
  def searchBatchLong(pats: Array[Array[Byte]]): Array[(Long, Long)] = {
    val k = pats.length
    val off = new Array[Long](k + 1)
    var q = 0
    while (q < k) { off(q + 1) = off(q) + pats(q).length; q += 1 }
    val flat = new Array[Byte](off(k).toInt)
    q = 0
    while (q < k) { System.arraycopy(pats(q), 0, flat, off(q).toInt, pats(q).length); q += 1 }
    val out = new Array[Long](2 * k)
    searchBatch0(h, flat, off, out)
    Array.tabulate(k)(j => (out(j), out(k + j)))
  }

  override def getPrevRange(sp: Int, ep: Int, c: Int): Option[(Int, Int)] = {
    val o = new Array[Long](2)
    prevRangeBatch0(h, Array(sp.toLong), Array(ep.toLong), Array(c.toByte), o)
    if (o(0) < o(1)) Some((o(0).toInt, o(1).toInt)) else None
  }

  /** getPrevRange for many (sp, ep, c) at once: (sp1, ep1) per triple, empty when sp1 >= ep1. */
  def prevRangeBatch(sp: Array[Long], ep: Array[Long], c: Array[Byte]): Array[(Long, Long)] = {
    val k = c.length
    val o = new Array[Long](2 * k)
    prevRangeBatch0(h, sp, ep, c, o)
    Array.tabulate(k)(j => (o(j), o(k + j)))
  }

  override def getIntervalPrevRange(sp: Int, ep: Int, cstart: Int, cend: Int): List[(Int, Int)] = {
    val o = new Array[Long](2 * math.max(cend - cstart + 1, 1))
    val k = intervalPrevRange0(h, sp, ep, cstart, cend, o)        // already in the reference's descending-c order (:47)
    List.tabulate(k)(j => (o(2 * j).toInt, o(2 * j + 1).toInt))
  }

  def close(): Unit = release()
  override def finalize(): Unit = release()
}

/** Same constructor arguments and sibling-file rule as NaiveFMSearcher (bwtmerger.scala:335-338, 17-36); X.fm is not
  * read (its content is a function of X.bwt and X.aux; `writeFm` produces it for the reference's own tools). */
class HipFMSearcher(filename: String, bigEndian: Boolean = true, device: Int = 0)
    extends SuffixWalkingAlgo with HipSuffixAlgo {
  import HipFM._
  protected val hbox =
    new AtomicLong(open0(BWTTempStorage.genBWTFilename(filename), BWTTempStorage.genAuxFilename(filename), bigEndian, device))
  val K = 256
  lazy val eof: Long = eof0(h)
  def handle: Long = h

  def nextSubstr(sp: Int, len: Int): String = fromLatin1(nextSubstr0(h, sp, len))
  def prevSubstr(sp: Int, len: Int): String = fromLatin1(prevSubstr0(h, sp, len))
  def getPrevI(i: Int): Int = { val (a, _) = prevRangeBatch(Array(i.toLong), Array(i + 1L), Array(bwtRead(i)))(0); a.toInt }
  def bwtRead(i: Int): Byte = prevSubstr0(h, i, 1)(0)          // BWTLoader.read: 0 at the EOF slot (bwtmerger.scala:155-162)

  /** nextSubstr for many rows in one launch: what rendering a result list needs (SAResult.toString, re2.scala:11-15). */
  def nextSubstrBatch(rows: Array[Long], len: Int): Array[String] = {
    val out = new Array[Byte](rows.length * len)
    val lens = new Array[Int](rows.length)
    nextSubstrBatch0(h, rows, len, out, lens)
    Array.tabulate(rows.length)(q => fromLatin1(out.slice(q * len, q * len + lens(q))))
  }

  /** FMCreator.create (bwtmerger.scala:424-533) from the device structure. */
  def writeFm(path: String): Unit = writeFm0(h, path)

  /** A batch in page-locked direct buffers (little-endian): the library moves it by DMA, pipelined against the
    * search.  `pat` = pattern bytes, `off` = k+1 longs, `out` receives sp[0..k) then ep[0..k) as longs. */
  def searchBatchDirect(pat: ByteBuffer, off: ByteBuffer, out: ByteBuffer, k: Long): Unit =
    searchBatchDirect0(h, pat, off, out, k)

  /** A batch of equal-length patterns (flat: k * len bytes) without an offsets array (fmx_search_batch_ex, fixed_len). */
  def searchBatchFixed(flat: Array[Byte], len: Int): Array[(Long, Long)] = {
    val k = flat.length / len
    val out = new Array[Long](2 * k)
    searchBatchFixed0(h, flat, len, out)
    Array.tabulate(k)(j => (out(j), out(k + j)))
  }

  /** The same over page-locked direct buffers with the intervals back in the 8-byte form (`out`: k + 1 + 2 * escapeCap
    * longs, little-endian; HipFMSearcher.unpack decodes): 40 bytes per 32-character pattern cross the link instead of 56. */
  def searchBatchPackedDirect(pat: ByteBuffer, len: Int, out: ByteBuffer, k: Long, escapeCap: Long): Unit =
    searchBatchPackedDirect0(h, pat, len, out, k, escapeCap)

  /** Build the derived tables now (fmx_prepare: 1 = k-mer table, 2 = select directory, 4 = the literal search's row tables, 8 = the regex frontier's) / free the row tables. */
  def prepare(what: Int): Unit = prepare0(h, what)
  /** The same under a budget: at most budgetBytes of device memory for all derived tables of this handle (fmx_prepare_ex). */
  def prepare(what: Int, budgetBytes: Long): Unit = prepareEx0(h, what, budgetBytes)
  /** This handle's own table policy (fmx_index_config_set): "ktab", "jump", "jump_pairs", "jump_chars", "tables_after", "table_budget". */
  def configSet(key: String, value: String): Unit = indexConfigSet0(h, key, value)
  def dropTables(): Unit = dropTables0(h, 4 | 8)
}

object HipFMSearcher {
  /** A page-locked direct buffer for batches (free with HipFM.hostFree0 when done). */
  def pinned(bytes: Long): ByteBuffer = HipFM.hostAlloc0(bytes).order(ByteOrder.LITTLE_ENDIAN)

  /** Decode of the 8-byte interval form (include/fmx.h): word q = sp | min(ep - sp, 0xFFFFFF) << 40, wider intervals in the
    * escape list behind word k (their count), as (q, ep) pairs.  Throws when more intervals were wide than the list holds. */
  def unpack(words: java.nio.LongBuffer, k: Int, escapeCap: Int): Array[(Long, Long)] = {
    val out = Array.tabulate(k) { q => val w = words.get(q); val sp = w & ((1L << 40) - 1); (sp, sp + (w >>> 40)) }
    val cnt = words.get(k)
    if (cnt > escapeCap) throw new Exception(cnt + " intervals of 2^24 - 1 rows or more, the escape list holds " + escapeCap)
    var j = 0
    while (j < cnt) {
      val ql = words.get(k + 1 + 2 * j)            // (checked as a Long: narrowing first would let 2^32 + 3 pass for pattern 3)
      if (ql < 0L || ql >= k.toLong) throw new Exception("escape entry " + j + " names pattern " + ql + " of " + k)      // fmx_unpack_intervals: FMX_ERR_FORMAT
      val q = ql.toInt
      out(q) = (out(q)._1, words.get(k + 2 + 2 * j))
      j += 1
    }
    out
  }
}

/** NaiveBWTSearcher(bwt, bucketStarts, rk0) (findex.scala:459-506): the searcher BWTMerger2.calcGaps uses. */
class HipBWTSearcher(bwt: Array[Byte], bucketStarts: Array[Long], rk0: Int, device: Int = 0) extends HipSuffixAlgo {
  protected val hbox = new AtomicLong(HipFM.openBlock0(bwt, bucketStarts, rk0, device))
  val K = bucketStarts.length
  // calcGaps (bwtmerger.scala:981-1023) asks ONE occ at a time, each depending on the last: answered on the host from
  // the library's copy of the dictionary (fmx_occ_host: a count + a scan of < 256 bytes), not by a kernel launch
  override def occ(c: Int, key: Int): Int = HipFM.occHost0(h, c & 0xff, key).toInt        // `val ci = c & 0xff`, :480

  /** The rank chain of calcGaps over text(from until text.length) from curRank = rank0 in one native call
    * (fmx_calc_gaps_chain): ranks(j) = curRank after byte from + j; returns how many bytes were processed -- fewer than
    * the rest when a rank equal to rklst needs kmpOut / longSuffixCmp (:1004-1010): fix that rank and call again. */
  def calcGapsChain(text: Array[Byte], from: Int, rank0: Long, lastChar: Int, rklst: Long, ranks: Array[Long]): Int =
    HipFM.calcGapsChain0(h, text, from, rank0, lastChar, rklst, ranks)
}

/** ReTree.matchSA on the device.  A regex string is parsed by the library with the reference's own grammar and tree
  * (REParser.re2post re2/re2.scala:50-185, ReTree.apply re2/retree.scala:156-370): "re2post syntax" and
  * scala.MatchError surface as Exceptions with those messages. */
object HipRegex {
  import HipFM._

  /** `ReTree(REParser.re2post(re)).matchSA(sa, maxBranching = .., maxIterations = ..)`: the reference's own queue
    * order and limits replayed on the device; the same list in the same order (newest first). */
  def matchSA(sa: HipFMSearcher, re: String, lineOnly: Boolean = false, maxBranching: Int = 1024,
              maxIterations: Int = 1000): List[SAResult] =
    matchBatch(sa, Array(re), lineOnly, MATCH_REFERENCE, 0, maxBranching, maxIterations)(0)

  /** Every match of every regex (the result multiset of matchSA whenever its limits do not bind), per regex sorted
    * by (len, sp, ep).  maxSteps bounds the match length (0 = 4096). */
  def matchAll(sa: HipFMSearcher, res: Array[String], lineOnly: Boolean = false, maxSteps: Int = 0): Array[List[SAResult]] =
    matchBatch(sa, res, lineOnly, MATCH_FRONTIER, maxSteps, 1024, 1000)

  def matchBatch(sa: HipFMSearcher, res: Array[String], lineOnly: Boolean, mode: Int, maxSteps: Int, maxBranching: Int,
                 maxIterations: Int, cap: Int = 1 << 20, maxFrontier: Long = 0): Array[List[SAResult]] = {
    // one native call compiles the whole batch on all host cores (fmx_regex_compile_batch); a regex the reference
    // would refuse raises what it raises there: "re2post syntax" (status 7) or a MatchError (status 8)
    // (the packed form is 0-terminated: a regex that contains the character 0 would be split in two silently -- the
    // one-regex entry point passes it through, here it is refused; the reference's readers escape byte 0 anyway)
    val zero = res.indexWhere(_.indexOf(0.toChar) >= 0)
    if (zero >= 0) throw new IllegalArgumentException("regex " + zero + " of the batch contains the character 0")
    val packed = res.flatMap(r => latin1(r) :+ 0.toByte)
    val handles = new Array[Long](res.length)
    val st = new Array[Int](res.length)
    regexCompileBatch0(packed, res.length, lineOnly, handles, st)
    val bad = st.indexWhere(_ != 0)
    if (bad >= 0) {
      regexFreeBatch0(handles)
      if (st(bad) == 7) throw new Exception("re2post syntax") else throw new MatchError(res(bad))
    }
    val batch = regexBatchCreate0(sa.handle, handles)
    try {
      val out = new Array[Long](3 * cap)
      val per = new Array[Int](res.length)
      val status = new Array[Int](1)
      val got = regexBatchMatch0(sa.handle, batch, Array(maxSteps, mode, maxBranching, maxIterations), maxFrontier, out,
                                 per, status).toInt
      val lists = Array.fill(res.length)(List[SAResult]())
      var j = got - 1
      while (j >= 0) {                                   // prepend from the end: each list keeps the library's order
        val rx = (out(3 * j) >>> 32).toInt
        lists(rx) ::= SAResult(sa, (out(3 * j) & 0xffffffffL).toInt, out(3 * j + 1).toInt, out(3 * j + 2).toInt)
        j -= 1
      }
      lists
    } finally {
      regexBatchFree0(batch)
      handles.foreach(regexFree0)
    }
  }
}

/** A batch of regexes kept resident on the device and matched many times (the serving shape of ReTree.matchSA over
  * many regexes).  Results land in page-locked direct buffers that the device writes itself: 24-byte records
  * (regex: Int, len: Int, sp: Long, ep: Long, little-endian), grouped by regex and sorted by (len, sp, ep), plus one
  * count per regex.  The buffers are reused by every call: read them before the next `matchRaw`. */
class HipRegexBatch(sa: HipFMSearcher, res: Array[String], lineOnly: Boolean = false, cap: Int = 1 << 22) {
  import HipFM._
  private val handles = res.map(r => regexCompile0(latin1(r), lineOnly))
  private val batch = regexBatchCreate0(sa.handle, handles)
  val results: ByteBuffer = HipFMSearcher.pinned(24L * cap)
  val perRegex: ByteBuffer = HipFMSearcher.pinned(4L * math.max(res.length, 1))
  private val status = new Array[Int](1)
  private val closed = new AtomicBoolean(false)
  var truncated = false

  /** Every match of every regex up to maxSteps characters (0 = 4096); returns the number of records in `results`. */
  def matchRaw(maxSteps: Int = 0, maxFrontier: Long = 0): Int = {
    val got = regexBatchMatchDirect0(sa.handle, batch, Array(maxSteps, MATCH_FRONTIER, 1024, 1000), maxFrontier, results,
                                     perRegex, status).toInt
    truncated = status(0) != 0
    got
  }

  /** The same as lists of SAResult per regex (ReTree.matchSA's result type). */
  def matchAll(maxSteps: Int = 0): Array[List[SAResult]] = {
    val got = matchRaw(maxSteps)
    val lists = Array.fill(res.length)(List[SAResult]())
    var j = got - 1
    while (j >= 0) {                                     // prepend from the end: each list keeps the library's order
      val at = 24 * j
      lists(results.getInt(at)) ::= SAResult(sa, results.getInt(at + 4), results.getLong(at + 8).toInt, results.getLong(at + 16).toInt)
      j -= 1
    }
    lists
  }

  def close(): Unit = if (closed.compareAndSet(false, true)) {      // once
    regexBatchFree0(batch)
    handles.foreach(regexFree0)
    hostFree0(results)
    hostFree0(perRegex)
  }
}
