"""COLMAP database writer over the standard library's sqlite3.

Same method names as reference vit_colmap/database/colmap_db.py:6-75, which wraps
`pycolmap.Database`; pycolmap is not assumed here, so `SqliteColmapDatabase` provides the subset
of its interface the reference touches (write_camera / write_image / write_keypoints /
write_descriptors / write_matches, num_* counters, exists_*), producing the on-disk layout COLMAP
reads [recalled schema, SURVEY.md §8b]:

  cameras(camera_id, model, width, height, params BLOB float64, prior_focal_length)
  images(image_id AUTOINCREMENT, name UNIQUE, camera_id)
  keypoints / descriptors(image_id, rows, cols, data BLOB float32 / uint8, row-major)
  matches(pair_id, rows, cols, data BLOB uint32), pair_id = id1 * 2147483647 + id2, id1 < id2
  two_view_geometries(pair_id, rows, cols, data BLOB uint32 inlier matches, config, F, E, H, qvec, tvec BLOB float64)
      — written by matching/two_view.py (geometric verification)
"""
import sqlite3
from contextlib import contextmanager
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

MAX_IMAGE_ID = 2147483647

CAMERA_MODEL_IDS = {"SIMPLE_PINHOLE": 0, "PINHOLE": 1, "SIMPLE_RADIAL": 2, "RADIAL": 3, "OPENCV": 4}
CAMERA_MODEL_NAMES = {v: k for k, v in CAMERA_MODEL_IDS.items()}

_SCHEMA = """
CREATE TABLE IF NOT EXISTS cameras (
    camera_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL,
    model INTEGER NOT NULL,
    width INTEGER NOT NULL,
    height INTEGER NOT NULL,
    params BLOB,
    prior_focal_length INTEGER NOT NULL);
CREATE TABLE IF NOT EXISTS images (
    image_id INTEGER PRIMARY KEY AUTOINCREMENT NOT NULL,
    name TEXT NOT NULL UNIQUE,
    camera_id INTEGER NOT NULL,
    CONSTRAINT image_id_check CHECK(image_id >= 0 and image_id < 2147483647),
    FOREIGN KEY(camera_id) REFERENCES cameras(camera_id));
CREATE TABLE IF NOT EXISTS keypoints (
    image_id INTEGER PRIMARY KEY NOT NULL,
    rows INTEGER NOT NULL,
    cols INTEGER NOT NULL,
    data BLOB,
    FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE);
CREATE TABLE IF NOT EXISTS descriptors (
    image_id INTEGER PRIMARY KEY NOT NULL,
    rows INTEGER NOT NULL,
    cols INTEGER NOT NULL,
    data BLOB,
    FOREIGN KEY(image_id) REFERENCES images(image_id) ON DELETE CASCADE);
CREATE TABLE IF NOT EXISTS matches (
    pair_id INTEGER PRIMARY KEY NOT NULL,
    rows INTEGER NOT NULL,
    cols INTEGER NOT NULL,
    data BLOB);
CREATE TABLE IF NOT EXISTS two_view_geometries (
    pair_id INTEGER PRIMARY KEY NOT NULL,
    rows INTEGER NOT NULL,
    cols INTEGER NOT NULL,
    data BLOB,
    config INTEGER NOT NULL,
    F BLOB,
    E BLOB,
    H BLOB,
    qvec BLOB,
    tvec BLOB);
CREATE UNIQUE INDEX IF NOT EXISTS index_name ON images(name);
"""


def pair_id_of(image_id1: int, image_id2: int) -> int:
    if image_id1 > image_id2:
        image_id1, image_id2 = image_id2, image_id1
    return image_id1 * MAX_IMAGE_ID + image_id2


def pair_id_to_image_ids(pair_id: int):
    image_id2 = pair_id % MAX_IMAGE_ID
    return (pair_id - image_id2) // MAX_IMAGE_ID, image_id2


def _quat_to_rot(q):
    w, x, y, z = np.asarray(q, np.float64) / max(np.linalg.norm(q), 1e-300)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _swap_two_view(m, F, E, H, qvec, tvec):
    """Two-view geometry of the pair (1 -> 2) expressed for (2 -> 1): x2^T F x1 = 0 <=> x1^T F^T x2 = 0 (same for E);
    x2 ~ H x1 <=> x1 ~ H^-1 x2 (a singular or all-zero H stays as it is: nothing to invert);
    X2 = R X1 + t <=> X1 = R^T X2 - R^T t, i.e. the conjugate quaternion and -R^T t."""
    m = np.asarray(m).reshape(-1, 2)[:, ::-1]
    F = None if F is None else np.asarray(F, np.float64).reshape(3, 3).T
    E = None if E is None else np.asarray(E, np.float64).reshape(3, 3).T
    if H is not None:
        H = np.asarray(H, np.float64).reshape(3, 3)
        if abs(np.linalg.det(H)) > 1e-300:
            H = np.linalg.inv(H)
    if qvec is not None:
        q = np.asarray(qvec, np.float64).reshape(4)
        t = np.zeros(3) if tvec is None else np.asarray(tvec, np.float64).reshape(3)
        tvec = -_quat_to_rot(q).T @ t
        qvec = q * np.array([1.0, -1.0, -1.0, -1.0])
    elif tvec is not None:
        tvec = -np.asarray(tvec, np.float64).reshape(3)
    return m, F, E, H, qvec, tvec


@dataclass
class Camera:
    """Stand-in for pycolmap.Camera(model=, width=, height=, params=) (vit_extractor.py:722-724)."""

    model: str = "SIMPLE_PINHOLE"
    width: int = 0
    height: int = 0
    params: list = field(default_factory=list)
    camera_id: Optional[int] = None
    has_prior_focal_length: bool = False


@dataclass
class Image:
    """Stand-in for pycolmap.Image(name=, camera_id=) (colmap_db.py:29)."""

    name: str = ""
    camera_id: int = 0
    image_id: Optional[int] = None


class SqliteColmapDatabase:
    """The slice of pycolmap.Database the reference uses."""

    def __init__(self, path: Optional[str] = None):
        self._conn = None
        if path is not None:
            self.open(path)

    # pycolmap 3.12: Database().open(path); 3.13: Database.open(path) static (colmap_db.py:8-16).
    # `SqliteColmapDatabase(path)` or `SqliteColmapDatabase().open(path)` both work here.
    def open(self, path: str):
        self._conn = sqlite3.connect(str(path))
        self._conn.executescript(_SCHEMA)
        self._conn.commit()
        return self

    def close(self):
        if self._conn is not None:
            self._conn.commit()
            self._conn.close()
            self._conn = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- writers -------------------------------------------------------------------------
    def write_camera(self, camera, use_camera_id: bool = False) -> int:
        model = camera.model if isinstance(camera.model, str) else getattr(camera.model, "name", str(camera.model))
        if model not in CAMERA_MODEL_IDS:
            raise ValueError(f"Unsupported camera model: {model}")
        params = np.asarray(camera.params, dtype=np.float64)
        cam_id = getattr(camera, "camera_id", None) if use_camera_id else None
        cur = self._conn.execute(
            "INSERT INTO cameras VALUES (?, ?, ?, ?, ?, ?)",
            (cam_id, CAMERA_MODEL_IDS[model], int(camera.width), int(camera.height), params.tobytes(),
             int(bool(getattr(camera, "has_prior_focal_length", False)))))
        self._conn.commit()
        return cur.lastrowid

    def write_image(self, image, use_image_id: bool = False) -> int:
        img_id = getattr(image, "image_id", None) if use_image_id else None
        cur = self._conn.execute("INSERT INTO images VALUES (?, ?, ?)", (img_id, image.name, int(image.camera_id)))
        self._conn.commit()
        return cur.lastrowid

    def _write_blob(self, table: str, key: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        rows, cols = (arr.shape[0], arr.shape[1]) if arr.ndim == 2 else (arr.shape[0], 1)
        self._conn.execute(f"INSERT OR REPLACE INTO {table} VALUES (?, ?, ?, ?)", (int(key), rows, cols, arr.tobytes()))
        self._conn.commit()

    def write_keypoints(self, image_id: int, keypoints: np.ndarray):
        self._write_blob("keypoints", image_id, np.asarray(keypoints, dtype=np.float32))

    def write_descriptors(self, image_id: int, descriptors: np.ndarray):
        self._write_blob("descriptors", image_id, np.asarray(descriptors, dtype=np.uint8))

    def write_matches(self, image_id1: int, image_id2: int, matches: np.ndarray, commit: bool = True):
        m = np.asarray(matches, dtype=np.uint32).reshape(-1, 2)
        if image_id1 > image_id2:  # stored relative to the smaller image id
            m = m[:, ::-1]
        m = np.ascontiguousarray(m)
        self._conn.execute("INSERT OR REPLACE INTO matches VALUES (?, ?, ?, ?)",
                           (pair_id_of(image_id1, image_id2), m.shape[0], 2, m.tobytes()))
        if commit:
            self._conn.commit()

    def write_two_view_geometry(self, image_id1: int, image_id2: int, inlier_matches: np.ndarray, config: int,
                                F=None, E=None, H=None, qvec=None, tvec=None, commit: bool = True):
        """One row per verified pair [recalled COLMAP layout]: inlier matches uint32 (rows, 2) relative to the smaller
        image id, `config` (TwoViewGeometry::ConfigurationType: 1 DEGENERATE, 2 CALIBRATED, 3 UNCALIBRATED, 4 PLANAR,
        5 PANORAMIC, 6 PLANAR_OR_PANORAMIC ... — the codes reference utils/metrics.py:131-141 names), and the
        matrices as float64 blobs (F, E, H row-major 3x3; qvec 4 as (w, x, y, z), tvec 3).  Geometry is stored for the
        direction smaller id -> larger id; a call with image_id1 > image_id2 is converted (`_swap_two_view`): match
        columns reversed, F and E transposed, H inverted, the relative pose inverted."""
        m = np.asarray(inlier_matches, dtype=np.uint32).reshape(-1, 2)
        if image_id1 > image_id2:
            m, F, E, H, qvec, tvec = _swap_two_view(m, F, E, H, qvec, tvec)
        m = np.ascontiguousarray(m)

        def blob(a, shape):
            a = np.zeros(shape) if a is None else np.asarray(a, np.float64).reshape(shape)
            return np.ascontiguousarray(a).tobytes()

        self._conn.execute(
            "INSERT OR REPLACE INTO two_view_geometries VALUES (?, ?, ?, ?, ?, ?, ?, ?, ?, ?)",
            (pair_id_of(image_id1, image_id2), m.shape[0], 2, m.tobytes() if m.shape[0] else None, int(config),
             blob(F, (3, 3)), blob(E, (3, 3)), blob(H, (3, 3)), blob(qvec if qvec is not None else [1, 0, 0, 0], (4,)),
             blob(tvec, (3,))))
        if commit:
            self._conn.commit()

    def commit(self):
        self._conn.commit()

    # ---- readers / counters ----------------------------------------------------------------
    def _count(self, sql: str) -> int:
        return int(self._conn.execute(sql).fetchone()[0])

    def num_cameras(self) -> int:
        return self._count("SELECT COUNT(*) FROM cameras")

    def num_images(self) -> int:
        return self._count("SELECT COUNT(*) FROM images")

    def num_keypoints(self) -> int:
        return self._count("SELECT COALESCE(SUM(rows), 0) FROM keypoints")

    def num_descriptors(self) -> int:
        return self._count("SELECT COALESCE(SUM(rows), 0) FROM descriptors")

    def num_matches(self) -> int:
        return self._count("SELECT COALESCE(SUM(rows), 0) FROM matches")

    def num_matched_image_pairs(self) -> int:
        return self._count("SELECT COUNT(*) FROM matches")

    def exists_keypoints(self, image_id: int) -> bool:
        return self._conn.execute("SELECT 1 FROM keypoints WHERE image_id = ?", (int(image_id),)).fetchone() is not None

    def exists_descriptors(self, image_id: int) -> bool:
        return self._conn.execute("SELECT 1 FROM descriptors WHERE image_id = ?", (int(image_id),)).fetchone() is not None

    def exists_matches(self, image_id1: int, image_id2: int) -> bool:
        return self._conn.execute("SELECT 1 FROM matches WHERE pair_id = ?",
                                  (pair_id_of(image_id1, image_id2),)).fetchone() is not None

    def read_all_images(self):
        return [Image(name=n, camera_id=c, image_id=i)
                for i, n, c in self._conn.execute("SELECT image_id, name, camera_id FROM images ORDER BY image_id")]

    def read_camera(self, camera_id: int) -> Camera:
        row = self._conn.execute("SELECT model, width, height, params, prior_focal_length FROM cameras "
                                 "WHERE camera_id = ?", (int(camera_id),)).fetchone()
        return Camera(model=CAMERA_MODEL_NAMES[row[0]], width=row[1], height=row[2],
                      params=list(np.frombuffer(row[3], dtype=np.float64)), camera_id=camera_id,
                      has_prior_focal_length=bool(row[4]))

    def _read_blob(self, table: str, key_col: str, key: int, dtype):
        row = self._conn.execute(f"SELECT rows, cols, data FROM {table} WHERE {key_col} = ?", (int(key),)).fetchone()
        if row is None:
            return None
        rows, cols, data = row
        if rows == 0 or data is None:
            return np.zeros((0, cols), dtype=dtype)
        return np.frombuffer(data, dtype=dtype).reshape(rows, cols).copy()

    def read_keypoints(self, image_id: int):
        return self._read_blob("keypoints", "image_id", image_id, np.float32)

    def read_descriptors(self, image_id: int):
        return self._read_blob("descriptors", "image_id", image_id, np.uint8)

    def num_verified_image_pairs(self) -> int:
        return self._count("SELECT COUNT(*) FROM two_view_geometries WHERE rows > 0")

    def num_inlier_matches(self) -> int:
        return self._count("SELECT COALESCE(SUM(rows), 0) FROM two_view_geometries")

    def read_two_view_geometry(self, image_id1: int, image_id2: int):
        """-> dict(inlier_matches uint32 (rows, 2), config, F, E, H, qvec, tvec) or None, in the direction
        image_id1 -> image_id2 (converted from the stored smaller -> larger direction when the ids are swapped)."""
        row = self._conn.execute("SELECT rows, cols, data, config, F, E, H, qvec, tvec FROM two_view_geometries WHERE pair_id = ?",
                                 (pair_id_of(image_id1, image_id2),)).fetchone()
        if row is None:
            return None
        rows, cols, data, config, F, E, H, qvec, tvec = row
        m = np.zeros((0, 2), np.uint32) if rows == 0 or data is None else np.frombuffer(data, np.uint32).reshape(rows, cols).copy()
        F, E, H = [np.frombuffer(x, np.float64).reshape(3, 3).copy() if x is not None else np.zeros((3, 3)) for x in (F, E, H)]
        qvec = np.frombuffer(qvec, np.float64).copy() if qvec is not None else np.array([1.0, 0, 0, 0])
        tvec = np.frombuffer(tvec, np.float64).copy() if tvec is not None else np.zeros(3)
        if image_id1 > image_id2:
            m, F, E, H, qvec, tvec = _swap_two_view(m, F, E, H, qvec, tvec)
        return dict(inlier_matches=np.ascontiguousarray(m), config=int(config), F=F, E=E, H=H, qvec=qvec, tvec=tvec)

    def read_matches(self, image_id1: int, image_id2: int):
        m = self._read_blob("matches", "pair_id", pair_id_of(image_id1, image_id2), np.uint32)
        if m is not None and image_id1 > image_id2:
            m = np.ascontiguousarray(m[:, ::-1])
        return m


class ColmapDatabase:
    """Reference vit_colmap/database/colmap_db.py:6-75 with `self.db` a SqliteColmapDatabase."""

    def __init__(self, db_path: str) -> None:
        self.db = SqliteColmapDatabase(str(db_path))

    # --- camera & image bookkeeping ---
    def add_pinhole_camera(self, width: int, height: int, fx: float, fy: float, cx: float, cy: float) -> int:
        return self.db.write_camera(Camera(model="PINHOLE", width=width, height=height, params=[fx, fy, cx, cy]))

    def add_image(self, name: str, camera_id: int) -> int:
        return self.db.write_image(Image(name=name, camera_id=camera_id))

    # --- features & matches ---
    def add_keypoints(self, image_id: int, kpts: np.ndarray) -> None:
        self.db.write_keypoints(image_id, kpts.astype(np.float32))

    def add_descriptors(self, image_id: int, desc: np.ndarray) -> None:
        self.db.write_descriptors(image_id, desc.astype(np.uint8))

    def add_matches(self, image_id1: int, image_id2: int, pairs: np.ndarray) -> None:
        self.db.write_matches(image_id1, image_id2, pairs.astype(np.uint32))

    def commit(self) -> None:
        self.db.commit()

    @staticmethod
    @contextmanager
    def open_database(db_path: str):
        db = SqliteColmapDatabase(str(db_path))
        try:
            yield db
        finally:
            db.close()

    @staticmethod
    def get_db_count(db, attr_name: str) -> int:
        """num_* are methods in pycolmap 3.13 and properties in 3.12 (colmap_db.py:67-75)."""
        attr = getattr(db, attr_name)
        return attr() if callable(attr) else attr
