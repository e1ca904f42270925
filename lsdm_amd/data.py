"""On-disk sample format of the reference's PROX-text dataset and a synthetic generator for it.

Restates the *format* consumed by ``posa.dataset.ProxDataset_txt`` (posa/dataset.py:348-474):

    <data_dir>/context/<seq>.txt           3 lines: prompt / space-separated given objects / target object
                                           (the third line is used unstripped, so files end without a newline, Q14)
    <data_dir>/reduced_vertices/<seq>.npy  (1024,3) human vertices
    <objs_dir>/<scene>/<cat>_<i>.npy       (1024,3) object point clouds; scene = <seq>.split('_')[0]

Iteration order: the reference enumerates ``os.listdir(context_dir)`` as the file system returns it (posa/dataset.py:374,503) --
unspecified and machine-dependent.  The loaders here default to ``order="sorted"`` (reproducible everywhere);
``order="listdir"`` replays the reference's enumeration on the same file system, which is what decides which draws of the
CLI's single RNG stream a sample receives (run/test_sdm.py:111,146: B = 1, one global generator).

``data_dir`` must contain an underscore-separated split name (``proxd_test``), as the reference parses
``data_dir.split('_')[1]`` (posa/dataset.py:369).  Each item is the tuple the sampling loop consumes:
(mask (9,), given_objs (9,1024,3), given_cats (9,C), target_obj (1024,3), target_cat (C,), prompt).

``HumaniseDataset`` restates ``posa.dataset.HUMANISE`` (posa/dataset.py:477-602), which differs in four places: its
own 11-entry category table (:538-552), the split taken from the last path component (``data_dir.split('/')[-1]``,
:497), the scene of a sequence being ``seq_name[:9] + '_00'`` (ScanNet ids, :580), and object files whose category is
not in the table being skipped (:567-572).
"""
import os
import zlib

import numpy as np
import torch
from torch.utils.data import Dataset

# category ids of posa/dataset.py:395-413 (two aliases share an id)
CATEGORY_IDS = {"human": 0, "chair": 1, "table": 2, "cabinet": 3, "sofa": 4, "bed": 5, "chest_of_drawers": 6, "chest": 6,
                "stool": 7, "tv_monitor": 8, "tv": 8, "lighting": 9, "shelving": 10, "seating": 11, "furniture": 12}


def _ordered(names, order):
    if order == "sorted":
        return sorted(names)
    if order == "listdir":
        return list(names)
    raise ValueError("order must be 'sorted' or 'listdir'")


class ProxTextDataset(Dataset):
    def __init__(self, data_dir, objs_data_dir='data/protext/objs', max_objs=8, pnt_size=1024, max_cats=13, order="sorted", **unused):
        self.data_dir, self.max_objs, self.pnt_size, self.max_cats = data_dir, max_objs, pnt_size, max_cats
        split = os.path.basename(os.path.normpath(data_dir)).split('_')
        if len(split) < 2:
            raise ValueError("data_dir must look like <name>_<split> (posa/dataset.py:369)")
        self.handle = 0 if split[1] == 'train' else 1
        ctx_dir = os.path.join(data_dir, "context")
        self.seq_names = _ordered([f[:-4] for f in os.listdir(ctx_dir) if f.endswith(".txt")], order)
        self.objs, self.cats = {}, {}
        for scene in os.listdir(objs_data_dir):
            self.objs[scene], self.cats[scene] = {}, {}
            for fn in os.listdir(os.path.join(objs_data_dir, scene)):
                name = fn[:-4]
                self.objs[scene][name] = np.load(os.path.join(objs_data_dir, scene, fn))
                self.cats[scene][name] = CATEGORY_IDS[name.split('.')[0].split('_')[0]]
        self.verts, self.context = {}, {}
        for s in self.seq_names:
            self.verts[s] = torch.tensor(np.load(os.path.join(data_dir, "reduced_vertices", s + ".npy")), dtype=torch.float32)
            with open(os.path.join(ctx_dir, s + ".txt")) as fh:
                prompt, given, target = fh.readlines()
            self.context[s] = (prompt.strip('\n'), given.strip('\n').split(' '), target)

    def __len__(self):
        return len(self.seq_names)

    def __getitem__(self, i):
        s = self.seq_names[i]
        scene = s.split('_')[0]
        prompt, given, target = self.context[s]
        verts = torch.zeros(self.max_objs + 1, self.pnt_size, 3)
        verts[0] = self.verts[s]
        mask = torch.zeros(self.max_objs + 1)          # the human slot keeps mask 0 (posa/dataset.py:458-464)
        cats = torch.zeros(self.max_objs + 1, self.max_cats)
        cats[0][CATEGORY_IDS['human']] = 1
        for k, obj in enumerate(given):
            verts[k + 1] = torch.tensor(self.objs[scene][obj], dtype=torch.float32)
            mask[k + 1] = 1
            cats[k + 1][CATEGORY_IDS[obj.split('_')[0]]] = 1
        target_cat = torch.zeros(self.max_cats)
        target_cat[CATEGORY_IDS[target.split('_')[0]]] = 1
        return mask, verts, cats, torch.tensor(self.objs[scene][target], dtype=torch.float32), target_cat, prompt


# category ids of posa/dataset.py:538-552
HUMANISE_CATEGORY_IDS = {"human": 0, "bed": 1, "sofa": 2, "table": 3, "door": 4, "desk": 5, "refrigerator": 6, "chair": 7,
                         "counter": 8, "bookshelf": 9, "cabinet": 10}


class HumaniseDataset(Dataset):
    """``posa.dataset.HUMANISE`` (posa/dataset.py:477-602): same item tuple as ``ProxTextDataset``, C = 11."""

    def __init__(self, data_dir, objs_data_dir='data/humanise/objs', max_objs=8, pnt_size=1024, max_cats=11, order="sorted", **unused):
        self.data_dir, self.max_objs, self.pnt_size, self.max_cats = data_dir, max_objs, pnt_size, max_cats
        self.handle = 0 if data_dir.split('/')[-1] == 'train' else 1
        ctx_dir = os.path.join(data_dir, "context")
        self.seq_names = _ordered([f.split('.txt')[0] for f in os.listdir(ctx_dir)], order)
        self.objs, self.cats = {}, {}
        for scene in os.listdir(objs_data_dir):
            self.objs[scene], self.cats[scene] = {}, {}
            for fn in os.listdir(os.path.join(objs_data_dir, scene)):
                name = fn[:-4]
                cat = name.split('_')[0]
                if cat in HUMANISE_CATEGORY_IDS:      # files of other categories are ignored (:567-572)
                    self.objs[scene][name] = np.load(os.path.join(objs_data_dir, scene, fn))
                    self.cats[scene][name] = HUMANISE_CATEGORY_IDS[cat]
        self.verts, self.context = {}, {}
        for s in self.seq_names:
            self.verts[s] = torch.tensor(np.load(os.path.join(data_dir, "reduced_vertices", s + ".npy")), dtype=torch.float32)
            with open(os.path.join(ctx_dir, s + ".txt")) as fh:
                prompt, given, target = fh.readlines()
            self.context[s] = (prompt.strip('\n'), given.strip('\n').split(' '), target)

    def __len__(self):
        return len(self.seq_names)

    def __getitem__(self, i):
        s = self.seq_names[i]
        scene = s[:9] + '_00'
        prompt, given, target = self.context[s]
        verts = torch.zeros(self.max_objs + 1, self.pnt_size, 3)
        verts[0] = self.verts[s]
        mask = torch.zeros(self.max_objs + 1)
        cats = torch.zeros(self.max_objs + 1, self.max_cats)
        cats[0][HUMANISE_CATEGORY_IDS['human']] = 1
        for k, obj in enumerate(given):
            verts[k + 1] = torch.tensor(self.objs[scene][obj], dtype=torch.float32)
            mask[k + 1] = 1
            cats[k + 1][HUMANISE_CATEGORY_IDS[obj.split('_')[0]]] = 1
        target_cat = torch.zeros(self.max_cats)
        target_cat[HUMANISE_CATEGORY_IDS[target.split('_')[0]]] = 1
        return mask, verts, cats, torch.tensor(self.objs[scene][target], dtype=torch.float32), target_cat, prompt


def write_synthetic_humanise(root, num_scenes=2, prompts_per_scene=2, seed=0, max_objs_per_sample=3):
    """Writes ``<root>/humanise/valid`` + ``<root>/humanise/objs`` in HUMANISE's layout: ScanNet-style scene ids
    (``scene0000_00``), sequences ``scene0000_<k>`` -- ``prompts_per_scene`` prompts per scene sharing the scene's objects
    (BASELINE configs[4]) -- and one object file of a category outside the table, which the loader must skip.
    Returns (data_dir, objs_dir)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    data_dir, objs_dir = os.path.join(root, "humanise", "valid"), os.path.join(root, "humanise", "objs")
    os.makedirs(os.path.join(data_dir, "context"), exist_ok=True)
    os.makedirs(os.path.join(data_dir, "reduced_vertices"), exist_ok=True)
    names = [k for k in HUMANISE_CATEGORY_IDS if k != "human"]
    for i in range(num_scenes):
        scene = "scene%04d_00" % i
        os.makedirs(os.path.join(objs_dir, scene), exist_ok=True)
        k = int(rng.integers(1, max_objs_per_sample + 1))
        objs = []
        for j in range(k + prompts_per_scene):
            cat = names[int(rng.integers(0, len(names)))]
            centre = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0, 1)])
            pts = (centre + rng.uniform(-1, 1, size=(1024, 3)) * rng.uniform(0.2, 0.8, size=3)).astype(np.float32)
            name = "%s_%d" % (cat, j)
            np.save(os.path.join(objs_dir, scene, name + ".npy"), pts)
            objs.append(name)
        np.save(os.path.join(objs_dir, scene, "window_99.npy"), np.zeros((1024, 3), np.float32))   # not in the table
        for q in range(prompts_per_scene):
            seq = "%s_%d" % (scene[:9], q)
            human = (rng.normal(0, 1, size=(1024, 3)) * np.array([0.25, 0.25, 0.45])).astype(np.float32)
            np.save(os.path.join(data_dir, "reduced_vertices", seq + ".npy"), human)
            target = objs[k + q]
            with open(os.path.join(data_dir, "context", seq + ".txt"), "w") as fh:
                fh.write("synthetic humanise prompt %d.%d: place the %s\n%s\n%s" % (i, q, target.split('_')[0], " ".join(objs[:k]), target))
    return data_dir, objs_dir


def write_synthetic_dataset(root, num_samples=4, seed=0, max_objs_per_sample=3):
    """Writes ``<root>/proxd_test`` + ``<root>/objs`` in the format above.  Returns (data_dir, objs_dir)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    data_dir, objs_dir = os.path.join(root, "proxd_test"), os.path.join(root, "objs")
    os.makedirs(os.path.join(data_dir, "context"), exist_ok=True)
    os.makedirs(os.path.join(data_dir, "reduced_vertices"), exist_ok=True)
    names = [k for k in CATEGORY_IDS if k not in ("human", "chest", "tv")]
    for i in range(num_samples):
        scene = "Scene%02d" % i
        seq = "%s_%05d_01" % (scene, i)
        os.makedirs(os.path.join(objs_dir, scene), exist_ok=True)
        k = int(rng.integers(1, max_objs_per_sample + 1))
        objs = []
        for j in range(k + 1):
            cat = names[int(rng.integers(0, len(names)))]
            centre = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0, 1)])
            pts = (centre + rng.uniform(-1, 1, size=(1024, 3)) * rng.uniform(0.2, 0.8, size=3)).astype(np.float32)
            name = "%s_%d" % (cat, j)
            np.save(os.path.join(objs_dir, scene, name + ".npy"), pts)
            objs.append(name)
        human = (rng.normal(0, 1, size=(1024, 3)) * np.array([0.25, 0.25, 0.45])).astype(np.float32)
        np.save(os.path.join(data_dir, "reduced_vertices", seq + ".npy"), human)
        with open(os.path.join(data_dir, "context", seq + ".txt"), "w") as fh:
            fh.write("synthetic prompt %d: place the %s\n%s\n%s" % (i, objs[-1].split('_')[0], " ".join(objs[:-1]), objs[-1]))
    return data_dir, objs_dir


def synthetic_text_features(prompts, dim=512):
    """Deterministic stand-in for CLIP features when the ``clip`` package/weights are unavailable (no network):
    N(0,1) draws seeded by the CRC32 of each prompt -- "random CLIP-text" of BASELINE.json."""
    out = np.stack([np.random.Generator(np.random.PCG64(zlib.crc32(p.encode()))).normal(0, 1, size=dim) for p in prompts])
    return torch.from_numpy(out.astype(np.float32))
