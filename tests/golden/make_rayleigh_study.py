#!/usr/bin/env python3
"""Parses the reference's only solver-produced numbers into tests/golden/rayleigh_study.json.

Source: /root/reference/OfflineBatchExamples/CompareRayleightWithFDTD/SummaryAnalysis.xlsx -- 309 water cases, each
comparing BabelViscoFDTD (Rayleigh source plane + FDTD) with the Rayleigh integral alone; metric definitions in
PART_2_AnalysisResults.ipynb cell 5, run recipe in PART_1_BabelBrain_RayleightTests.ipynb. The workbook is read as a
zip of XML (no openpyxl in the image). Data only: case names and their metric values.

Run:  python tests/golden/make_rayleigh_study.py
"""
import json
import os
import re
import xml.etree.ElementTree as ET
import zipfile

SRC = '/root/reference/OfflineBatchExamples/CompareRayleightWithFDTD/SummaryAnalysis.xlsx'
HERE = os.path.dirname(os.path.abspath(__file__))
NS = '{http://schemas.openxmlformats.org/spreadsheetml/2006/main}'


def read_rows(path):
    z = zipfile.ZipFile(path)
    shared = []
    if 'xl/sharedStrings.xml' in z.namelist():
        for si in ET.fromstring(z.read('xl/sharedStrings.xml')).findall(NS + 'si'):
            shared.append(''.join(t.text or '' for t in si.iter(NS + 't')))
    rows = []
    for r in ET.fromstring(z.read('xl/worksheets/sheet1.xml')).iter(NS + 'row'):
        row = []
        for c in r.findall(NS + 'c'):
            v, t = c.find(NS + 'v'), c.get('t')
            if v is None:
                inline = c.find(NS + 'is')
                row.append(''.join(x.text or '' for x in inline.iter(NS + 't')) if inline is not None else None)
            elif t == 's':
                row.append(shared[int(v.text)])
            else:
                row.append(float(v.text))
        rows.append(row)
    return rows


def main():
    rows = read_rows(SRC)
    head = rows[0]
    cases = []
    for r in rows[1:]:
        d = dict(zip(head, r))
        d['case'] = int(d['case'])
        desc = d['Description']
        m = re.match(r'ZAdj_(-?[\d.]+)_DEEP_Single_(\d+)kHz_(\d+)PPW_Foc([\d.]+)_Diam([\d.]+)\.nii\.gz', desc)
        if m:
            d['tx'] = 'Single'
            d['zadj_mm'], d['freq_khz'], d['ppw'], d['focal_mm'], d['diam_mm'] = (float(m.group(1)), int(m.group(2)), int(m.group(3)),
                                                                                 float(m.group(4)), float(m.group(5)))
        else:
            d['tx'] = 'CTX_500' if 'CTX_500' in desc else ('H317' if 'H317' in desc else 'REMOPD')
        d['L Inf location'] = [int(x) for x in re.findall(r'-?\d+', d['L Inf location'])]
        cases.append(d)
    out = {'source': 'OfflineBatchExamples/CompareRayleightWithFDTD/SummaryAnalysis.xlsx', 'columns': head, 'cases': cases}
    json.dump(out, open(os.path.join(HERE, 'rayleigh_study.json'), 'w'), indent=0)
    print('wrote %d cases (%d Single)' % (len(cases), sum(c['tx'] == 'Single' for c in cases)))
    # element centres of the H317 array (data table of the reference: 128 rows "Element,X,Y,Z" in inches, H317.py:17-21),
    # stored in metres with z measured up from the apex plane like H317Locations does (z = F - Z)
    rows = [l.strip().split(',') for l in open('/root/reference/TranscranialModeling/H-317 XYZ Coordinates_revB update 1.18.22.csv')][1:]
    xyz = [[float(r[1]) * 25.4e-3, float(r[2]) * 25.4e-3, 135e-3 - float(r[3]) * 25.4e-3] for r in rows if len(r) == 4]
    assert len(xyz) == 128
    json.dump({'source': 'TranscranialModeling/H-317 XYZ Coordinates_revB update 1.18.22.csv', 'focal_m': 135e-3, 'element_diameter_m': 9.5e-3,
               'centres_m': xyz}, open(os.path.join(HERE, 'h317_elements.json'), 'w'))
    print('wrote 128 H317 element centres')
    # element centres of the REMOPD flat array (data table of the reference: REMOPD_ElementPosition.mat, 256 x 3, metres)
    from scipy.io import loadmat
    pos = loadmat('/root/reference/TranscranialModeling/REMOPD_ElementPosition.mat')['REMOPD_ElementPosition']
    assert pos.shape == (256, 3)
    json.dump({'source': 'TranscranialModeling/REMOPD_ElementPosition.mat', 'pitch_m': 3.08e-3, 'kerf_m': 0.5e-3, 'aperture_m': 0.058,
               'centres_m': [[float(v) for v in r] for r in pos]}, open(os.path.join(HERE, 'remopd_elements.json'), 'w'))
    print('wrote 256 REMOPD element centres')


if __name__ == '__main__':
    main()
