"""FASTA header -> id rules of PhaMers' scripts/id_parser.py (host string handling that
count_file needs on either side of the GPU path)."""


def represents_float(s):
    try:
        float(s)
        return True
    except (TypeError, ValueError):
        return False


def get_contig_id(header):
    """scripts/id_parser.py:18-30: the field after 'ID' in an '_'-separated header."""
    header = header.strip().replace('>', '')
    parts = header.split('_')
    return parts[1 + parts.index('ID')].replace('-circular', '')


def is_genbank_id(id):
    """scripts/id_parser.py:80-86."""
    return not represents_float(id) and len(id) >= 2 and id[-2] == '.'


def get_bacteria_id(header):
    """scripts/id_parser.py:57-68."""
    id = header.split(' ')[0]
    if is_genbank_id(id):
        return id
    id = header.split('\t')[1].replace('>', '')
    if is_genbank_id(id):
        return id


def get_phage_id(header):
    """scripts/id_parser.py:71-77."""
    return header.split('|')[3].replace('>', '')


def get_id(header):
    """scripts/id_parser.py:89-100."""
    if '_ID_' in header:
        return get_contig_id(header)
    elif header.count('|') == 4:
        return get_phage_id(header)
    else:
        return get_bacteria_id(header)
